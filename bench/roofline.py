"""The `roofline` object of the bench line: the dominant kernel family is the decode weight stream (k_gemv / k_gemv32 / k_gemv32g: every LM
linear + the tied LM head, 145 launches per decode step, HBM-bound).

  achieved = algorithmic bytes per launch / average launch duration
  algorithmic bytes per launch = (5.550 GB of layer linears + 0.622 GB of LM head) / 145 = 42.56 MB  (fp8: layer linears halved)
  average launch duration      = measured IN SITU, in this run: a rocprofv3 --kernel-trace child pass over a short static decode of the same
                                 engine configuration (every k_gemv* dispatch inside the hipGraph-replayed decode steps, attention and norm
                                 launches between them as in production) -- the figure the committed profiles/r06_bench_*_kernel_stats.md reproduce.
                                 The launch-only replay of rounds 1-5 (HIP events around the 145 launches back to back) stays as a side field.
  traffic  = HBM bytes per launch from two more child passes (--pmc FETCH_SIZE / WRITE_SIZE, kernel trace only beside them)
  decode_step_frac = bytes one decode step must stream (weights once + the KV cache of the batch) / the measured step / 8 TB/s: the number that
                     moves tiles/s -- it also pays for the attention, norm and bookkeeping launches between the weight streams.
"""
from __future__ import annotations

import ctypes as C
import glob
import json
import os
import shutil
import sqlite3
import subprocess
import tempfile

import torch

from .common import HBM_PEAK_GBS, KV_BYTES_PER_TOKEN, N_NEW, PYTHON, ROOT, child_env, lm_weight_bytes

P = lambda x: C.c_void_p(x.data_ptr())      # noqa: E731


def gemv_replay_us(wl, MB, fp8):
    """Average duration of the decode step's 145 weight-streaming launches at batch MB when they are replayed back to back (HIP events on the launch
    stream; distinct weight copies per layer so that nothing is served from L2 / Infinity Cache).  Launch-only: no attention / norm launches between."""
    from socioreasoner_amd import lib as L
    lib, dev, t_ = L.load(), wl.dev, wl.geom.text
    H, QN, I = t_.hidden_size, (t_.num_attention_heads + 2 * t_.num_key_value_heads) * t_.head_dim, t_.intermediate_size
    nl, V = t_.num_hidden_layers, t_.vocab_size
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    bf = dict(dtype=torch.bfloat16, device=dev)
    wq, wo = torch.empty(nl, QN, H, **bf).normal_(0, 0.02), torch.empty(nl, H, H, **bf).normal_(0, 0.02)
    wg, wd = torch.empty(nl, 2 * I, H, **bf).normal_(0, 0.02), torch.empty(nl, H, I, **bf).normal_(0, 0.02)
    wv = torch.empty(V, H, **bf).normal_(0, 0.02)
    if fp8:            # fp8 images + scales of the four layer linears (values do not matter for the timing)
        w8 = [torch.empty(nl, n_ * k_, dtype=torch.uint8, device=dev).random_(0, 120) for n_, k_ in ((QN, H), (H, H), (2 * I, H), (H, I))]
        sc8 = torch.ones(2 * I, dtype=torch.float32, device=dev)
    # batches > 4: activations travel between the launches in fragment order, exactly as in the engine's decode layer (engine.hip
    # enqueue_decode_forward: x_tiled / out_tiled); buffers hold whole 16-row groups
    XT, OT, TL = (0x800, 0x1000, 0x100) if MB > 4 else (0, 0, 0x100)
    Mp = (MB + 15) // 16 * 16
    x, act = torch.empty(Mp, I, **bf).normal_(0, 1), torch.empty(Mp, I, **bf).normal_(0, 1)
    part = torch.empty(4, MB, QN, dtype=torch.float32, device=dev)
    lg = torch.empty(MB, V, dtype=torch.float32, device=dev)
    nw, bq = torch.ones(H, **bf), torch.zeros(QN, **bf)
    slabs = torch.zeros(2, MB, H, dtype=torch.float32, device=dev)
    xo, xr, qkv_o = torch.zeros(MB, H, **bf), torch.zeros(MB, H, **bf), torch.empty(MB, QN, **bf)
    fused = MB <= 4         # same launch configuration as the engine's decode layer
    nb = lib.sr_op_gemv_f32_blocks(V, MB, H, 1 if fused else 0)
    av, ai = torch.empty(MB, nb, dtype=torch.float32, device=dev), torch.empty(MB, nb, dtype=torch.int32, device=dev)
    eps, ksd = C.c_float(1e-6), 4 if MB > 16 else 2
    NW = P(nw) if fused else None
    SL, NS, XO = (P(slabs), 2, P(xo)) if fused else (None, 0, None)

    def seq():
        for l in range(nl):
            if fp8:
                lib.sr_op_gemv_f8(P(x), I, P(w8[0][l]), P(sc8), MB, QN, H, P(qkv_o), QN, 3, P(bq), NW, eps, 1, s)
                lib.sr_op_gemv_f8(P(x), I, P(w8[1][l]), P(sc8), MB, H, H, P(xr), H, 4, None, None, eps, 1, s)
                lib.sr_op_gemv_f8(P(x), I, P(w8[2][l]), P(sc8), MB, 2 * I, H, P(act), I, 1, None, NW, eps, 1, s)
                lib.sr_op_gemv_f8(P(act), I, P(w8[3][l]), P(sc8), MB, H, I, P(part), H, 0, None, None, eps, ksd, s)
                continue
            lib.sr_op_gemv_fused(P(x), I, P(wq[l]), MB, QN, H, P(qkv_o), QN, 3 | TL | XT, P(bq), NW, eps, SL, NS, XO, None, None, s)
            lib.sr_op_gemv_fused(P(x), I, P(wo[l]), MB, H, H, P(xr), H, 4 | TL | XT, None, None, eps, None, 0, None, None, None, s)
            lib.sr_op_gemv_fused(P(x), I, P(wg[l]), MB, 2 * I, H, P(act), I, 1 | TL | XT | OT, None, NW, eps, None, 0, None, None, None, s)
            lib.sr_op_gemv(P(act), I, P(wd[l]), MB, H, I, P(part), ksd, 0 | TL | XT, s)
        lib.sr_op_gemv_fused(P(x), I, P(wv), MB, V, H, P(lg), V, 2 | TL | XT, None, NW, eps, SL, NS, XO, P(av), P(ai), s)
    seq()
    a, b_ = wl.ev(), wl.ev()
    reps = 5
    a.record()
    for _ in range(reps):
        seq()
    b_.record()
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b_) / reps / (4 * nl + 1) * 1e3


def _rocprof(cmd_tail, counters=None, timeout_s=240):
    """Run `rocprofv3 --kernel-trace [--pmc ...] -- <cmd_tail>` as a child process (cwd /tmp, this process's GPU idle meanwhile); returns the path
    of the rocpd database inside a fresh temporary directory (the caller removes it), or (None, None)."""
    if shutil.which("rocprofv3") is None:
        return None, None
    td = tempfile.mkdtemp(prefix="sr_prof_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace"] + (["--pmc", counters] if counters else []) + ["-d", td, "-o", "p", "--"] + cmd_tail
    try:
        subprocess.run(cmd, cwd="/tmp", env=child_env(TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
    except Exception:  # noqa: BLE001  (a profiler that crashes or times out must never take the bench line with it)
        shutil.rmtree(td, ignore_errors=True)
        return None, None
    dbs = glob.glob(os.path.join(td, "**", "*.db"), recursive=True)
    if not dbs:
        shutil.rmtree(td, ignore_errors=True)
        return None, None
    return dbs[0], td


def measure_gemv_in_situ(args, B, steps=1):
    """The decode weight stream timed where it runs: a rocprofv3 kernel trace of `bench.py --static --batch B` (this script, one static batch per
    step, hipGraph-replayed decode, nothing else) as a child process.  Returns {"avg_us": launch-count-weighted average duration of every k_gemv*
    dispatch, "launches": n, "per_kernel_avg_us": {...}, "other_decode_kernels_avg_us": {...}} or None (no rocprofv3 / a failed pass)."""
    cmd = [PYTHON, os.path.join(ROOT, "bench.py"), "--static", "--batch", str(B), "--steps", str(steps), "--warmup", "1", "--no-latency",
           "--no-cpu-baseline", "--no-pmc", "--tile", str(args.tile)]
    cmd += (["--fp8-mx"] if args.fp8_mx else ["--fp8"] if args.fp8 else []) + (["--pair"] if args.pair else [])
    db, td = _rocprof(cmd)
    if db is None:
        return None
    try:
        rows = sqlite3.connect(db).execute(
            "select s.kernel_name, count(*), sum(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
            "group by s.kernel_name").fetchall()
    except Exception:  # noqa: BLE001
        return None
    finally:
        shutil.rmtree(td, ignore_errors=True)
    short = lambda n: n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "").split("(")[0][:44]      # noqa: E731
    gemv = [(n, c, t) for n, c, t in rows if "k_gemv" in n]
    calls = sum(c for _, c, _ in gemv)
    if not calls:
        return None
    other = [(n, c, t) for n, c, t in rows if any(k in n for k in ("k_rmsnorm_row", "k_attn_dec", "k_step"))]
    return {"avg_us": round(sum(t for _, _, t in gemv) / calls / 1e3, 3), "launches": calls,
            "per_kernel_avg_us": {short(n): round(t / c / 1e3, 2) for n, c, t in sorted(gemv, key=lambda r: -r[2])},
            "other_decode_kernels_avg_us": {short(n): round(t / c / 1e3, 2) for n, c, t in sorted(other, key=lambda r: -r[2])}}


def measure_gemv_traffic(fp8):
    """HBM bytes of the decode weight-stream launches measured IN THIS RUN: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes
    (kernel trace only beside them, as gpurun requires) over tools/probe_r2.py -- the batch-32 and batch-1 launches of this kernel family on
    weight-sized operands -- reduced by tools/rocpd_pmc.py / tools/gemv_traffic.py (FETCH_SIZE doubled per the gfx950 note of the MI355X guide).
    Returns the ratio dict, or None (no rocprofv3 on the box, a failed pass)."""
    outs, tds = {}, []
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            db, td = _rocprof([PYTHON, os.path.join(ROOT, "tools", "probe_r2.py"), "gemv"] + (["fp8"] if fp8 else []), counters=c, timeout_s=120)
            if db is None:
                return None
            tds.append(td)
            outs[c] = os.path.join(td, c + ".json")
            subprocess.run([PYTHON, os.path.join(ROOT, "tools", "rocpd_pmc.py"), db, outs[c]], capture_output=True, text=True, timeout=120, check=True)
        res = os.path.join(tds[0], "traffic.json")
        subprocess.run([PYTHON, os.path.join(ROOT, "tools", "gemv_traffic.py"), outs["FETCH_SIZE"], outs["WRITE_SIZE"], res] + (["fp8"] if fp8 else []),
                       capture_output=True, text=True, timeout=120, check=True)
        j = json.load(open(res))
        return j if j.get("traffic_over_algorithmic_weighted_batch32") else None
    except Exception:  # noqa: BLE001
        return None
    finally:
        for td in tds:
            shutil.rmtree(td, ignore_errors=True)


def _committed_ratio(fp8, key):
    f = os.path.join(ROOT, "profiles", "r05_pmc_gemv_traffic_fp8.json" if fp8 else "r05_pmc_gemv_traffic_bf16.json")
    try:
        return json.load(open(f)).get(key), os.path.relpath(f, ROOT)
    except Exception:  # noqa: BLE001
        return None, None


def build(wl, decode_step_ms, latency=None, measure=True):
    """The roofline object of the headline configuration (and latency["roofline"] for the batch-1 leg when it ran).  `decode_step_ms`: one decode
    step that had the chip to itself, as measured in the timed region.  measure=False (--no-pmc, N > 1, --no-latency): no child passes."""
    a, B, fp8 = wl.args, wl.B, wl.args.fp8
    nl = wl.geom.text.num_hidden_layers
    n_launch = 4 * nl + 1
    wl_bytes, wh_bytes = lm_weight_bytes(wl.geom)
    if fp8:
        wl_bytes = wl_bytes / 2          # the layer linears stream 1 byte per weight (+ 4 bytes per output channel, < 0.1 %)
    bpl = (wl_bytes + wh_bytes) / n_launch
    key = lambda mb: "traffic_over_algorithmic_weighted_batch32" if mb > 4 else "traffic_over_algorithmic_weighted_batch1"      # noqa: E731
    live = None
    if measure:
        torch.cuda.empty_cache()
        live = measure_gemv_traffic(fp8)

    def one(mb, step_ms, s_ctx):
        replay = gemv_replay_us(wl, mb, fp8)
        situ = measure_gemv_in_situ(a, mb, steps=1 if mb > 1 else 2) if measure else None
        us = situ["avg_us"] if situ else replay
        ratio, src = None, None
        if live:
            ratio = live.get(key(mb))
            src = ("measured IN THIS RUN: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two child passes over tools/probe_r2.py gemv; FETCH_SIZE "
                   "doubled per the gfx950 note): HBM bytes / algorithmic bytes x bytes_per_launch")
        if ratio is None:
            ratio, f = _committed_ratio(fp8, key(mb))
            src = f"{f}: committed ratio x bytes_per_launch -- NOT measured in this run" if ratio else None
        step_bytes = wl_bytes + wh_bytes + KV_BYTES_PER_TOKEN * (s_ctx + N_NEW / 2) * mb
        situ_src = ("IN SITU, this run: rocprofv3 --kernel-trace child pass over `bench.py --static --batch %d`: every k_gemv* dispatch inside the "
                    "graph-replayed decode steps" % mb)
        replay_src = ("launch-only replay (HIP events around the 145 launches back to back), this run -- no rocprofv3 child pass "
                      "(--no-pmc / N > 1 / unavailable)")
        r = {"bound": "hbm",
             "kernel": f"k_gemv family at batch {mb} (decode weight stream: all LM linears + LM head)" + (" [fp8 layer linears]" if fp8 else ""),
             "achieved": round(bpl / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bpl / us / 1e3 / HBM_PEAK_GBS, 4),
             "traffic": round(ratio * bpl) if ratio else None, "traffic_over_algorithmic": ratio, "traffic_source": src,
             "bytes_per_launch": round(bpl), "launches_per_decode_step": n_launch, "avg_launch_us": round(us, 2),
             "avg_launch_us_source": situ_src if situ else replay_src,
             "avg_launch_us_replay": round(replay, 2), "in_situ": situ,
             "decode_step_ms": round(step_ms, 4) if step_ms and step_ms > 0 else None}
        if step_ms and step_ms > 0:
            r["decode_step_bytes"] = round(step_bytes)
            r["decode_step_achieved_GBs"] = round(step_bytes / (step_ms * 1e-3) / 1e9, 1)
            r["decode_step_frac"] = round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        return r
    roof = one(B, decode_step_ms, wl.s_prompt)
    if latency is not None:
        latency["roofline"] = one(1, latency["decode_step_ms"], wl.s_prompt)
    return roof
