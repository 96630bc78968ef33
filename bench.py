#!/usr/bin/env python3
"""Benchmark of the SocioReasoner-3B inference hot path on MI355X (contract: see the round prompt / DESIGN.md).

One "step" = one pass of the hot path over one batch of synthetic tiles per GPU:
  uint8 448x448 tile -> patchify -> ViT (1024 patches) -> merger (256 tokens) -> LM prefill (448-token prompt)
  -> greedy decode of exactly 128 tokens (EOS ignored) -> raster tail (union of 4 756^2 masks -> nearest 768^2 -> IoU).

Default workload (the headline ``value``) = BASELINE.json configs[2]: SocioReasoner-3B bf16, one MI355X, 32 rows in flight,
CONTINUOUS BATCHING (admit on finish): every step serves 64 tile requests through 32 batch rows.  The same command also
times configs[1] (batch 1, the latency configuration) and reports it under ``latency_b1`` of the same JSON line.
``--batch 1`` makes configs[1] the headline instead; ``--static`` replaces the scheduler by one static batch per step.
Inputs (tiles, masks, weights) are resident in HBM before timing.
Multi-GPU: one process per GPU (torchrun), tiles sharded data-parallel, no data-path collective while generating,
one RCCL all-gather of the per-tile results (tokens + IoU counts) per step -> weak scaling.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16
# algorithmic work per tile
VIT_GFLOP, PREFILL_GFLOP = 1342.9, 2516.3
N_NEW = 128
RAGGED_LO, RAGGED_HI = 64, 192      # ragged phase: per-request max_new uniform in [64, 192] (mean 128)
GRID = (1, 32, 32)


def lm_weight_bytes(g):
    t = g.text
    qn = (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
    per_layer = (qn * t.hidden_size + t.hidden_size * t.num_attention_heads * t.head_dim + 3 * t.intermediate_size * t.hidden_size) * 2
    return per_layer * t.num_hidden_layers, t.vocab_size * t.hidden_size * 2


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks ourselves -- one process per GPU under
    torch.distributed.run on 127.0.0.1 (the reference fans a batch out over its DP workers the same way:
    /root/reference/roll/distributed/scheduler/decorator.py:106-181) -- and return their exit code; None = this process is a rank
    (or N == 1) and runs the bench itself.  Refuses (exit code 2) when the node has fewer GPUs than ranks, unless
    SR_DIST_BACKEND=gloo asks for the host-staged development layout in which ranks share devices."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and os.environ.get("SR_DIST_BACKEND") != "gloo":
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs (one rank per GPU over RCCL), this node has {have}; "
              f"set SR_DIST_BACKEND=gloo to let ranks share devices on a development box", file=sys.stderr, flush=True)
        return 2
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # N schedulers + N poll loops share this host: give every rank its share of the cores for its intra-op pool (torch.distributed.run would set
    # OMP_NUM_THREADS=1; the collator and the PNG writers of the pipeline use a few threads), never more than 8
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or args.gpus) // args.gpus))))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="batch rows per GPU (32 = configs[2], the default; 1 = configs[1])")
    ap.add_argument("--static", action="store_true", help="one static batch of --batch tiles per step instead of continuous batching")
    ap.add_argument("--continuous", action="store_true", help="(default for --batch > 1) serve --waves x batch requests through the scheduler")
    ap.add_argument("--tile", type=int, default=448, choices=[448, 896], help="tile edge in pixels (896 = BASELINE.json configs[4]'s high-res tiles: 4096 patches, "
                    "1024 image tokens, 1216-token prompt); the MFMA fractions are only quoted for 448")
    ap.add_argument("--pair", action="store_true", help="reference-faithful sample: TWO images (map + satellite tile) per request, 706-token prompt "
                    "(SURVEY.md section 8(D) 'reported separately'); value is then samples/s")
    ap.add_argument("--waves", type=int, default=4, help="continuous mode: a step serves waves x batch tile requests through the batch rows")
    ap.add_argument("--admit-cus", default="auto", help="CUs per shader engine (of 8) given to the overlapped admission stream, or auto (chosen per admission)")
    ap.add_argument("--no-overlap", action="store_true", help="continuous mode: admit between decode steps on one stream (round-1 behaviour) instead of "
                    "staging the next admission on a CU-masked stream under the running rows' decode")
    ap.add_argument("--drain", action="store_true", help="continuous mode: drain the batch rows between steps (every step starts with an exposed admission on an idle "
                    "engine, rounds 1-3) instead of serving the steps' requests as ONE stream (step k + 1's first admission staged under step k's last rows)")
    ap.add_argument("--poll", type=int, default=16, help="continuous mode: decode steps queued per scheduling round (rows are released / admitted between rounds)")
    ap.add_argument("--poll-ragged", type=int, default=4, help="the same for the ragged phase (0 = --poll): rows that end on different steps are refilled sooner with short "
                    "rounds -- measured 59.8 / 60.8 / 62.1 tiles/s at 16 / 8 / 4")
    ap.add_argument("--no-latency", action="store_true", help="skip the additional batch-1 (configs[1]) measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sam", action="store_true", help="skip the SAM2 (seg_infer) timing")
    ap.add_argument("--no-more-rows", action="store_true", help="skip the 64- and 128-row points (child processes)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the two-stage pipeline timing with SAM2 at work (a child process)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure the weight stream's HBM traffic with rocprofv3 in this run (two child passes); use the ratio committed under profiles/")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="BASELINE.json configs[4] weights: LM decoder linears fp8 e4m3, per-channel scale")
    ap.add_argument("--fp8-mx", action="store_true", help="--fp8 plus MX fp8 activations in the prefill linears (fp8 x fp8 block-scaled MFMA, lm_weight_dtype 2)")
    ap.add_argument("--gather-logits", action="store_true",
                    help="verification mode (static batch): all-gather the float32 logits of every decode step (north_star's literal exchange)")
    args = ap.parse_args()
    rc = self_launch(args)
    if rc is not None:
        sys.exit(rc)
    if args.fp8_mx:
        args.fp8 = True
    B = args.batch
    continuous = (B > 1 and not args.static and not args.gather_logits) or args.continuous
    if args.gather_logits or args.no_graph:
        continuous = False

    overlap = continuous and not args.no_overlap

    from socioreasoner_amd import dp, hostops, raster, synthetic
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    from socioreasoner_amd import lib as L

    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("SR_DIST_BACKEND", "nccl") == "nccl":
        # let RCCL describe its communicator (rank count, transport of every channel) to a per-rank file the exchange report parses
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
        os.environ["SR_RCCL_LOG"] = os.environ["NCCL_DEBUG_FILE"] = f"/tmp/sr_rccl_{os.getpid()}_rank{os.environ.get('RANK', '0')}.log"
    rank, world, local = dp.init_distributed()      # RCCL when ranks > 1 (gloo only if SR_DIST_BACKEND=gloo asks for it)
    if world > 1 and os.environ.get("OMP_NUM_THREADS", "").isdigit():
        torch.set_num_threads(max(1, int(os.environ["OMP_NUM_THREADS"])))      # the per-rank share of the host cores (self_launch)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s): launch `python bench.py --gpus N` (self-launching) or "
                         f"torchrun --nproc-per-node N bench.py --gpus N")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    geom = geometry_3b()
    global GRID, VIT_GFLOP, PREFILL_GFLOP
    GRID = (1, args.tile // 14, args.tile // 14)
    NPATCH = GRID[1] * GRID[2]
    NIMG = 2 if args.pair else 1
    S_PROMPT = 96 + 94 + NIMG * (2 + NPATCH // 4)
    if args.tile != 448 or args.pair:
        VIT_GFLOP = PREFILL_GFLOP = float("nan")       # (the constants above are the 448-tile counts)
    eng = Engine(geom, max_patches=NPATCH * NIMG * B, max_prefill_tokens=S_PROMPT * B, max_batch=B, max_ctx=max(640, (S_PROMPT + RAGGED_HI + 63) // 64 * 64), max_new_tokens=RAGGED_HI, device=str(dev), lm_fp8="mx" if args.fp8_mx else args.fp8,
                 kv_slots=2 * B if overlap else 0)      # spare KV slots: the next requests are prefilled while the current rows decode
    t0 = time.time()
    eng.load_synthetic_weights(seed=0)
    load_s = time.time() - t0

    # ---- synthetic inputs, resident in HBM.  Request k of a step is tile (rank * n_req + k)
    n_req = args.waves * B if continuous else B
    tiles = [rank * n_req + i for i in range(n_req)]
    imgs = [[torch.from_numpy(synthetic.tile_pixels(NIMG * i + j, args.tile, args.tile)).to(dev) for j in range(NIMG)] for i in tiles]
    ids = [synthetic.tile_prompt(geom, i, GRID, n_images=NIMG) for i in tiles]
    pos3 = []
    for x in ids:
        p, _ = hostops.get_rope_index(torch.from_numpy(x)[None], [GRID] * NIMG, None, image_token_id=geom.image_token_id,
                                      vision_start_token_id=geom.vision_start_token_id)
        pos3.append(p[:, 0].numpy())
    mk = [synthetic.tile_masks(i) for i in tiles]
    masks = torch.from_numpy(__import__("numpy").stack([m for m, _ in mk], axis=1)).to(dev).contiguous()      # [4, n_req, 756, 756]
    gts = torch.from_numpy(__import__("numpy").stack([g for _, g in mk], axis=0)).to(dev).contiguous()         # [n_req, 768, 768]
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def raster_tail(lo, n):
        """union of the 4 object masks of tiles lo..lo+n-1 -> nearest 756 -> 768 -> IoU counts vs the ground truth.  The tiles
        are stacked along the row axis: 756 / 768 = 63 / 64 is exact in binary, so the nearest-row rule of the stack equals the
        per-tile rule (floor(b * 756 + y * 63 / 64) = b * 756 + floor(y * 63 / 64)) and one launch serves all tiles."""
        acc = torch.zeros(n * 756, 756, dtype=torch.uint8, device=dev)
        for j in range(4):
            raster.mask_union_(acc, masks[j, lo:lo + n].reshape(n * 756, 756))
        up = raster.resize_nearest(acc, n * 768, 768).reshape(n, 768, 768)
        return raster.iou_counts_batched(up, gts[lo:lo + n])          # one launch for the n tiles

    def steps_continuous(k_steps, phase_ms=None):
        """k_steps steps of waves x B requests each through B rows, served as ONE request stream: the later requests are admitted as rows
        free up (EOS is ignored by the metric, so all rows of a wave finish together; the point is the measured cost of the request-level
        path), step k + 1's first group is staged under step k's last rows like any other group; a step's raster tail and result exchange
        run when its last request completes.  --drain: one scheduler per step on an idle engine (rounds 1-3)."""
        from socioreasoner_amd.serving import ContinuousBatcher, Request
        res = None
        groups = [[s_] for s_ in range(k_steps)] if args.drain else [list(range(k_steps))]
        for grp in groups:
            cb = ContinuousBatcher(eng, eos=[], pad_id=0, steps_per_poll=args.poll, time_phases=phase_ms is not None, overlap=overlap, admit_cus_per_se=args.admit_cus if args.admit_cus == "auto" else float(args.admit_cus))
            reqs = [Request(ids=ids[k], pos3=pos3[k], max_new=N_NEW, images=imgs[k], grids=[GRID] * NIMG, tag=(s_, k)) for s_ in grp for k in range(n_req)]
            toks = {s_: [None] * n_req for s_ in grp}
            left = {s_: n_req for s_ in grp}
            spans, out = [], {}

            def finished(req, t):
                s_, k = req.tag
                toks[s_][k] = t
                left[s_] -= 1
                if left[s_] == 0:           # the step's last request: its raster tail + the one exchange of the step (every rank gets here in step order)
                    e0, e1 = ev(), ev()
                    e0.record()
                    counts = raster_tail(0, n_req)
                    e1.record()
                    spans.append((e0, e1))
                    r_ = torch.cat([torch.tensor(toks[s_], dtype=torch.int64, device=dev), counts], dim=1)
                    out[s_] = dp.all_gather_rows(r_, n_req * world) if world > 1 else r_
            cb.run_stream(reqs, finished)
            res = out[grp[-1]]
            if phase_ms is not None:
                for k, v in cb.phase_ms().items():
                    phase_ms[k] = phase_ms.get(k, 0.0) + v
                for k in ("admitted", "staged_shared", "steps", "steps_shared", "rounds", "host_ms", "poll_wait_ms"):
                    sched[k] = sched.get(k, 0) + cb.stats[k]
                shares.extend(cb.stats["shares"])
                sched["share_model"] = cb.stats.get("share_model")
                phase_ms["raster"] += sum(a.elapsed_time(b_) for a, b_ in spans)
        return res

    def step_static(nb, phase_ms=None, first=0, gather=True):
        e0, e1, e2, e3, e4 = ev(), ev(), ev(), ev(), ev()
        e0.record()
        pix = torch.cat([eng.patchify(im) for grp in imgs[first:first + nb] for im in grp], dim=0)
        emb = eng.vit_forward(pix, [GRID] * (nb * NIMG))
        e1.record()
        first_logits = eng.prefill(ids[first:first + nb], pos3[first:first + nb], emb, return_logits=args.gather_logits)
        e2.record()
        if args.gather_logits:
            alltoks, bad = dp.decode_with_logits_gather(lambda t: eng.decode_step(t), first_logits, N_NEW, nb * world)
            assert bad == 0, f"{bad} on-device argmax results differ from the argmax of the gathered logits"
            toks = alltoks[rank * nb:(rank + 1) * nb]
        else:
            toks = eng.decode(N_NEW, use_graph=not args.no_graph)
        e3.record()
        counts = raster_tail(first, nb)
        e4.record()
        res = torch.cat([toks.to(torch.int64), counts], dim=1)             # [nb, 130] per-tile result row
        if world > 1 and nb == B and gather:
            res = dp.all_gather_rows(res, nb * world)                      # the one RCCL exchange of the step
        if phase_ms is not None:
            torch.cuda.synchronize(dev)
            for k, a, b_ in (("vit", e0, e1), ("prefill", e1, e2), ("decode", e2, e3), ("raster", e3, e4)):
                phase_ms[k] += a.elapsed_time(b_)
        return res

    phase_ms = {"vit": 0.0, "prefill": 0.0, "decode": 0.0, "raster": 0.0}
    shares = []         # CU share (of 8 per shader engine) of every overlapped admission
    sched = {}          # continuous mode: requests admitted / staged under decode, decode steps alone / sharing the chip
    def run_steps(k_steps, rec=False):
        if continuous:
            return steps_continuous(k_steps, phase_ms if rec else None)
        r_ = None
        for _ in range(k_steps):
            r_ = step_static(B, phase_ms if rec else None)
        return r_

    if world > 1:      # open the RCCL communicator outside the timed region even with --warmup 0
        dp.all_gather_rows(torch.zeros(B, 1, dtype=torch.int64, device=dev), B * world)
    if args.warmup:
        run_steps(args.warmup)
    dp.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    res = run_steps(args.steps, True)
    torch.cuda.synchronize(dev)
    dp.barrier()
    dt = time.perf_counter() - t0
    dt = dp.all_reduce_max(dt, dev)
    tiles_per_s = world * n_req * args.steps / dt
    exchange = dp.exchange_info()
    if world > 1:
        # the N-rank record verifies itself: the communicator really has N ranks (hard error otherwise -- a scaling number from a
        # degenerate group would be worthless), and what RCCL logged about its transports is reported as checks, not guessed
        assert exchange["nranks"] == world == args.gpus, exchange
        if exchange.get("backend") == "nccl":
            exchange["checks"] = {"rccl_logged_nranks_eq_world": exchange.get("log_nranks") == [world],
                                  "channels_via_p2p_xgmi": exchange.get("channels_via_p2p", 0) > 0,
                                  "no_channel_via_net": exchange.get("channels_via_net", 0) == 0,
                                  "no_host_staging": True}      # all_gather_into_tensor on device buffers (dp._gather_equal)
            exchange["verified"] = all(exchange["checks"].values())

    # ---- the same kernels as ONE static batch of B tiles (no scheduler, whole chip, warm): the phase times the MFMA fractions are quoted on
    static_ref = None
    if rank == 0 and continuous and not args.no_latency:
        st_ms = {"vit": 0.0, "prefill": 0.0, "decode": 0.0, "raster": 0.0}
        step_static(B, gather=False)          # (rank 0 only: no collective in here)
        for _ in range(2):
            step_static(B, st_ms, gather=False)
        fw = (st_ms["vit"] + st_ms["prefill"]) / 2
        static_ref = {"workload": f"one static batch of {B} tiles per step (same engine, same kernels, no scheduler)", "steps": 2,
                      "phase_ms": {k: round(v / 2, 3) for k, v in st_ms.items()},
                      "decode_step_ms": round(st_ms["decode"] / 2 / (N_NEW - 1), 4),
                      "vit_mfma_frac": round(VIT_GFLOP * B / (st_ms["vit"] / 2 * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4),
                      "prefill_mfma_frac": round(PREFILL_GFLOP * B / (st_ms["prefill"] / 2 * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4),
                      "forward_mfma_frac": round((VIT_GFLOP + PREFILL_GFLOP) * B / (fw * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4)}

    # ---- the same step on a DRAINED engine (rounds 1-3's definition of a step: every step starts with an exposed admission on idle rows)
    drained = None
    if rank == 0 and world == 1 and continuous and not args.drain and not args.no_latency:
        args.drain = True
        n_dr = 2                               # (the engine, its graphs and the scheduler's calibration are warm from the timed region above)
        torch.cuda.synchronize(dev)
        t_ = time.perf_counter()
        steps_continuous(n_dr)
        torch.cuda.synchronize(dev)
        d_ = (time.perf_counter() - t_) / n_dr
        args.drain = False
        drained = {"workload": f"rounds 1-3's definition of a step, {n_dr} of them back to back: every step starts on an idle engine (its first admission is exposed) and its last rows "
                               "decode with nothing staged under them", "steps": n_dr, "tiles_per_s": round(n_req / d_, 3), "ms_per_step": round(d_ * 1e3, 2)}

    # ---- beyond the headline's 32 rows (not the headline: BASELINE.json configs[2] says batch = 32): the same workload through 64 and 128 batch
    # rows per GPU (the reference's request-level mode keeps up to 128 requests in flight per worker, generate_scheduler.py:57).  The decode
    # GEMVs stream every weight tile once for all rows (k_gemv32g), so rows per step grow faster than the step.  Each point is this
    # script run as a child process (own engine, 2 steps of 2 x rows requests, no side measurements).
    more_rows = None
    if rank == 0 and world == 1 and continuous and B == 32 and args.tile == 448 and not args.pair and not args.fp8 and not args.no_latency and not args.no_more_rows:
        import subprocess
        more_rows = {}
        for rows_ in (64, 128):
            cmd = [sys.executable, os.path.abspath(__file__), "--batch", str(rows_), "--steps", "2", "--warmup", "1", "--waves", "2", "--no-latency", "--no-cpu-baseline", "--no-sam"]
            try:
                r_ = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
                j_ = json.loads([ln for ln in r_.stdout.splitlines() if ln.startswith("{")][-1])
                more_rows[str(rows_)] = {"tiles_per_s": j_["value"], "ms_per_step": j_["ms_per_step"], "tiles_per_step": j_["config"]["tiles_per_gpu_per_step"],
                                         "decode_step_ms_alone": j_["roofline"]["decode_step_ms"], "decode_step_ms_shared": j_["phase_ms_per_step"]["scheduler"]["decode_step_ms_shared"],
                                         "gemv_avg_launch_us": j_["roofline"]["avg_launch_us"], "gemv_hbm_frac": j_["roofline"]["frac"],
                                         "forward_mfma_frac": j_["phase_ms_per_step"]["forward_mfma_frac"], "vit_mfma_frac": j_["phase_ms_per_step"]["vit_mfma_frac"],
                                         "workspace_GB": j_["workspace_GB"]}
            except Exception as e_:  # noqa: BLE001
                more_rows[str(rows_)] = {"error": f"{type(e_).__name__}: {e_}"[:300]}

    # ---- admit-on-finish TIMED: the same requests with ragged answer lengths (per-request max_new uniform in [64, 192], mean 128, seeded),
    # through the same scheduler, against static batches of B that each run to their longest answer.  The headline's rows all stop on
    # the same step (EOS is ignored by the metric), so only this phase shows what refilling rows as they free up is worth.
    ragged = None
    if rank == 0 and world == 1 and continuous and not args.no_latency:      # (single-GPU side measurements: N > 1 runs go straight to the line)
        import numpy as _np
        from socioreasoner_amd.serving import ContinuousBatcher, Request
        lens = _np.random.default_rng(4000).integers(RAGGED_LO, RAGGED_HI + 1, n_req).tolist()

        def ragged_run(ov):
            cb = ContinuousBatcher(eng, eos=[], pad_id=0, steps_per_poll=args.poll_ragged or args.poll, overlap=ov,
                                   admit_cus_per_se=args.admit_cus if args.admit_cus == "auto" else float(args.admit_cus))
            reqs = [Request(ids=ids[k], pos3=pos3[k], max_new=int(lens[k]), images=imgs[k], grids=[GRID] * NIMG) for k in range(n_req)]
            torch.cuda.synchronize(dev)
            t_ = time.perf_counter()
            toks_ = cb.run(reqs)
            torch.cuda.synchronize(dev)
            dt_ = time.perf_counter() - t_
            assert [len(t) for t in toks_] == lens
            return dt_, cb.stats["steps"]

        ragged_run(overlap)                                   # warm (graphs, calibration)
        r_dt, r_steps = ragged_run(overlap)
        # static batches: B requests at a time, every batch decodes until its longest answer is done
        torch.cuda.synchronize(dev)
        t_ = time.perf_counter()
        s_steps = 0
        for lo in range(0, n_req, B):
            pix = torch.cat([eng.patchify(im) for grp in imgs[lo:lo + B] for im in grp], dim=0)
            emb = eng.vit_forward(pix, [GRID] * (min(B, n_req - lo) * NIMG))
            eng.prefill(ids[lo:lo + B], pos3[lo:lo + B], emb)
            eng.decode(max(lens[lo:lo + B]))
            s_steps += max(lens[lo:lo + B])
        torch.cuda.synchronize(dev)
        s_dt = time.perf_counter() - t_
        ragged = {"workload": f"{n_req} requests, max_new uniform in [{RAGGED_LO}, {RAGGED_HI}] (mean {sum(lens) / len(lens):.1f}, seed 4000), {B} rows, {args.poll_ragged or args.poll} decode steps per scheduling round",
                  "continuous_tiles_per_s": round(n_req / r_dt, 3), "continuous_tokens_per_s": round(sum(lens) / r_dt, 1), "continuous_decode_steps": r_steps,
                  "static_batches_tiles_per_s": round(n_req / s_dt, 3), "static_batches_decode_steps": s_steps,
                  "gain": round(s_dt / r_dt, 4)}

    # ---- SAM2 (Hiera-L) behind seg_infer: the mask half of a tile in the reference's pipeline (seg_strategy.py:47-60) -- 756 x 756 image ->
    # set_image, then decode + arg-max + resize + OR per object.  Timed beside the LM path (the metric's tile uses synthetic masks: SURVEY 8(D)).
    sam = None
    if rank == 0 and world == 1 and not args.no_latency and not args.no_sam:
        from socioreasoner_amd import sam2 as _sam2
        sg = _sam2.Sam2Geometry()
        simg = torch.from_numpy(synthetic.tile_pixels(0, 756, 756)).to(dev)
        simgs = [torch.from_numpy(synthetic.tile_pixels(i, 756, 756)).to(dev) for i in range(8)]
        sobj = [dict(point_coords=[[300 + 20 * k, 320]], point_labels=[1], box=[100 + 30 * k, 120, 420 + 30 * k, 600]) for k in range(4)]
        ssd = _sam2.synthetic_state_dict(sg)

        def sam_mode(dtype):
            se = _sam2.Sam2Engine(sg, str(dev), dtype=dtype)
            se.load_state_dict(ssd)
            sacc = torch.zeros(756, 756, dtype=torch.uint8, device=dev)

            def sam_tile():                      # the reference's loop: one image, one object at a time
                se.set_image(simg)
                for o in sobj:
                    se.predict_or(sacc, **o)

            def sam_tiles8():                    # 8 tiles per encoder pass, the 4 objects of a tile per decoder pass
                se.set_images(simgs)
                for b_ in range(8):
                    se.select(b_)
                    se.predict_or_many(sacc, sobj)
            sam_tile()
            sam_tiles8()
            t_ = {}
            for nm, fn, reps in (("set_image_ms", lambda: se.set_image(simg), 5), ("set_images_8_ms", lambda: se.set_images(simgs), 3),
                                 ("predict_ms_per_object", lambda: se.predict_or(sacc, **sobj[0]), 20), ("predict_ms_4_objects_one_pass", lambda: se.predict_or_many(sacc, sobj), 20),
                                 ("tile_ms_4_objects", sam_tile, 5), ("tiles8_ms_4_objects", sam_tiles8, 3)):
                torch.cuda.synchronize(dev)
                t0_ = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize(dev)
                t_[nm] = round((time.perf_counter() - t0_) / reps * 1e3, 3)
            del se
            torch.cuda.empty_cache()
            return dict(t_, tiles_per_s_4_objects=round(1e3 / t_["tile_ms_4_objects"], 2), tiles_per_s_4_objects_batched=round(8e3 / t_["tiles8_ms_4_objects"], 2))
        # Hiera-L encoder: 1.57 TFLOP of Linear layers + 0.21 TFLOP of attention per 1024 x 1024 input
        f32 = sam_mode(torch.float32)
        f32["dtype"] = ("float32 (the reference's precision: seg_infer's default).  Round 5: the Linear layers run on the bf16 matrix pipe from an exact three-term bf16 split of "
                        "both operands (six partial products, float32 accumulation: csrc/sam_f32.hip k_gemm_f32s); attention on v_mfma_f32_16x16x4_f32")
        f32["encoder_TFLOPs_batched"] = round(1.78e12 * 8 / (f32["set_images_8_ms"] * 1e-3) / 1e12, 1)
        f32["encoder_vs_f32_mfma_peak_157TF"] = round(1.78e12 * 8 / (f32["set_images_8_ms"] * 1e-3) / 157.3e12, 4)
        b16 = sam_mode(torch.bfloat16)
        b16["dtype"] = "bf16 storage / float32 accumulation (opt-in: sam2_compute_dtype bf16; masks differ from float32's inside the bf16 noise band)"
        sam = {"workload": "SAM2 Hiera-L (216.9 M parameters, random init), 756 x 756 tiles -> 1024 x 1024 input, box + click prompts, 3 masks + scores per object, "
                           "best mask resized to 756 x 756 and OR-ed on the device; tile_ms = one tile and one object at a time (the reference's loop), "
                           "tiles8_ms = 8 tiles per encoder pass and a tile's 4 objects per decoder pass (what seg_infer runs)",
               "float32": f32, "bf16": b16}

    # ---- the reference's own two-stage pipeline (examples/infer/rlvr_megatron.yaml through SocioSegInferPipeline: two generate calls on a
    # (map, satellite) pair each, two segment calls, four PNGs + two text files per sample) with SAM2 DOING WORK: random weights emit no <answer>,
    # so tools/run_example_small.py scripts the decoded answers (4 objects per stage; the LM still generates its 128 tokens per stage on the engine).
    # A child process (own engines), after this process's side measurements; phase wall times from SocioSegInferPipeline.timing.
    pipeline = None
    if rank == 0 and world == 1 and continuous and B == 32 and args.tile == 448 and not args.pair and not args.fp8 and not args.no_latency and not args.no_pipeline:
        import subprocess
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        env.update(SCRIPTED_OBJECTS="4", SOCIOSEG_NUM_SAMPLES="64", NEW_TOKENS=str(N_NEW), OUT="/tmp/sr_bench_pipeline_out")
        try:
            r_ = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_example_small.py")], capture_output=True, text=True, timeout=900, env=env)
            pipeline = json.loads([ln for ln in r_.stdout.splitlines() if ln.startswith("{")][-1])
            pipeline["workload"] = ("SocioSegInferPipeline.run() on 64 synthetic SocioSeg samples, the shipped YAML (3B LM + SAM2 Hiera-L float32, synthetic weights), 128 new tokens per "
                                    "stage, decoded answers scripted to 4 objects per stage so that seg_infer encodes every satellite image and decodes 4 prompts per stage and sample")
        except Exception as e_:  # noqa: BLE001
            pipeline = {"error": f"{type(e_).__name__}: {e_}"[:300]}

    # ---- configs[1] beside it: one tile at a time on the same engine (batch-1 kernels, hipGraph decode), rank 0 only
    latency = None
    if rank == 0 and world == 1 and B > 1 and not args.no_latency and not args.fp8:
        lat_ms = {"vit": 0.0, "prefill": 0.0, "decode": 0.0, "raster": 0.0}
        step_static(1)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        ksteps = max(args.steps, 3)
        for _ in range(ksteps):
            step_static(1, lat_ms)
        torch.cuda.synchronize(dev)
        d1 = time.perf_counter() - t1
        # opt-in split-K of the small-M residual GEMMs (engine.hip prefill_splitk: off by default because it gives up bit-exact batch
        # invariance): one extra batch-1 step with it switched on, reported beside the default numbers
        sk_ms = {"vit": 0.0, "prefill": 0.0, "decode": 0.0, "raster": 0.0}
        os.environ["SR_SPLITK"] = "1"
        L.reload_switches()            # the library reads its switches once, not per call
        step_static(1)
        step_static(1, sk_ms)
        os.environ.pop("SR_SPLITK")
        L.reload_switches()
        latency = {"workload": "BASELINE.json configs[1]: batch 1, one tile per step", "tiles_per_s": round(ksteps / d1, 4),
                   "ms_per_tile": round(d1 / ksteps * 1e3, 3), "steps": ksteps,
                   "phase_ms": {k: round(v / ksteps, 3) for k, v in lat_ms.items()},
                   "decode_step_ms": round(lat_ms["decode"] / ksteps / (N_NEW - 1), 4),
                   "vit_mfma_frac": round(VIT_GFLOP / (lat_ms["vit"] / ksteps * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4),
                   "prefill_mfma_frac": round(PREFILL_GFLOP / (lat_ms["prefill"] / ksteps * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4),
                   "opt_in_splitk": {"prefill_ms": round(sk_ms["prefill"], 3),
                                     "prefill_mfma_frac": round(PREFILL_GFLOP / (sk_ms["prefill"] * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4),
                                     "note": "SR_SPLITK=1: o_proj / down-projection of prefills <= 1024 rows split over K; not the default (float32 association differs from the batched kernels)"}}

    # ---- roofline of the dominant kernel (k_gemv, the decode weight stream): HIP events on the launch stream around
    # the exact per-step launch sequence of that kernel (145 launches: 4 per layer + LM head) on weight-sized operands
    if rank == 0:
        lib = L.load()
        t_ = geom.text
        H, QN, I = t_.hidden_size, (t_.num_attention_heads + 2 * t_.num_key_value_heads) * t_.head_dim, t_.intermediate_size
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        P = lambda x: C.c_void_p(x.data_ptr())
        nl = t_.num_hidden_layers
        # distinct weight copies per layer so that nothing is served from L2 / Infinity Cache (256 MB) between launches
        wq = torch.empty(nl, QN, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
        wo = torch.empty(nl, H, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
        wg = torch.empty(nl, 2 * I, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
        wd = torch.empty(nl, H, I, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
        wv = torch.empty(t_.vocab_size, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
        if args.fp8:            # fp8 images + scales of the four layer linears (values do not matter for the timing)
            w8 = [torch.empty(nl, n_ * k_, dtype=torch.uint8, device=dev).random_(0, 120) for n_, k_ in ((QN, H), (H, H), (2 * I, H), (H, I))]
            sc8 = torch.ones(2 * I, dtype=torch.float32, device=dev)
        n_launch = 4 * nl + 1
        wl, wh = lm_weight_bytes(geom)
        if args.fp8:
            wl = wl / 2          # the layer linears stream 1 byte per weight (+ 4 bytes per output channel, < 0.1 %)
        bytes_per_launch = (wl + wh) / n_launch

        def gemv_roofline(MB):
            """average launch duration of the decode step's weight-streaming launches at batch MB (events on the launch stream)"""
            # batches > 4: activations travel between the launches in fragment order, exactly as in the engine's decode layer
            # (engine.hip enqueue_decode_forward: x_tiled / out_tiled); buffers hold whole 16-row groups
            XT = 0x800 if MB > 4 else 0
            OT = 0x1000 if MB > 4 else 0
            Mp = (MB + 15) // 16 * 16
            x = torch.empty(Mp, I, dtype=torch.bfloat16, device=dev).normal_(0, 1)
            part = torch.empty(4, MB, QN, dtype=torch.float32, device=dev)
            act = torch.empty(Mp, I, dtype=torch.bfloat16, device=dev).normal_(0, 1)
            lg = torch.empty(MB, t_.vocab_size, dtype=torch.float32, device=dev)
            nw = torch.ones(H, dtype=torch.bfloat16, device=dev)
            bq = torch.zeros(QN, dtype=torch.bfloat16, device=dev)
            slabs = torch.zeros(2, MB, H, dtype=torch.float32, device=dev)
            xo = torch.zeros(MB, H, dtype=torch.bfloat16, device=dev)
            qkv_o = torch.empty(MB, QN, dtype=torch.bfloat16, device=dev)
            xr = torch.zeros(MB, H, dtype=torch.bfloat16, device=dev)
            fused = MB <= 4         # same launch configuration as the engine's decode layer (engine.hip enqueue_decode_forward)
            nb = lib.sr_op_gemv_f32_blocks(t_.vocab_size, MB, H, 1 if fused else 0)
            av = torch.empty(MB, nb, dtype=torch.float32, device=dev)
            ai = torch.empty(MB, nb, dtype=torch.int32, device=dev)
            TL = 0x100              # weights are fragment-ordered in the engine; the timing does not depend on the values
            eps = C.c_float(1e-6)
            ksd = 4 if MB > 16 else 2

            def seq():
                for l in range(nl):
                    if args.fp8:
                        lib.sr_op_gemv_f8(P(x), I, P(w8[0][l]), P(sc8), MB, QN, H, P(qkv_o), QN, 3, P(bq), P(nw) if fused else None, eps, 1, s)
                        lib.sr_op_gemv_f8(P(x), I, P(w8[1][l]), P(sc8), MB, H, H, P(xr), H, 4, None, None, eps, 1, s)
                        lib.sr_op_gemv_f8(P(x), I, P(w8[2][l]), P(sc8), MB, 2 * I, H, P(act), I, 1, None, P(nw) if fused else None, eps, 1, s)
                        lib.sr_op_gemv_f8(P(act), I, P(w8[3][l]), P(sc8), MB, H, I, P(part), H, 0, None, None, eps, ksd, s)
                        continue
                    lib.sr_op_gemv_fused(P(x), I, P(wq[l]), MB, QN, H, P(qkv_o), QN, 3 | TL | XT, P(bq), P(nw) if fused else None, eps,
                                         P(slabs) if fused else None, 2 if fused else 0, P(xo) if fused else None, None, None, s)
                    lib.sr_op_gemv_fused(P(x), I, P(wo[l]), MB, H, H, P(xr), H, 4 | TL | XT, None, None, eps, None, 0, None, None, None, s)
                    lib.sr_op_gemv_fused(P(x), I, P(wg[l]), MB, 2 * I, H, P(act), I, 1 | TL | XT | OT, None, P(nw) if fused else None, eps, None, 0, None, None, None, s)
                    lib.sr_op_gemv(P(act), I, P(wd[l]), MB, H, I, P(part), ksd, 0 | TL | XT, s)
                lib.sr_op_gemv_fused(P(x), I, P(wv), MB, t_.vocab_size, H, P(lg), t_.vocab_size, 2 | TL | XT, None, P(nw) if fused else None, eps,
                                     P(slabs) if fused else None, 2 if fused else 0, P(xo) if fused else None, P(av), P(ai), s)
            seq()
            a, b_ = ev(), ev()
            reps = 5
            a.record()
            for _ in range(reps):
                seq()
            b_.record()
            torch.cuda.synchronize(dev)
            return a.elapsed_time(b_) / reps / n_launch

        avg_ms = gemv_roofline(B)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        steps_total = args.steps * (N_NEW - 1)
        # continuous mode: the decode steps that had the chip to themselves (steps that shared it with an admission are listed apart)
        decode_step_ms = phase_ms["decode"] / steps_total if not continuous else phase_ms["decode"] / max(sched["steps"] - sched["steps_shared"], 1)
        kv_bytes = 36864.0 * (S_PROMPT + N_NEW / 2) * B
        # HBM traffic per launch from the PMC passes committed under profiles/ (FETCH_SIZE doubled per the gfx950 note):
        # measured ratio traffic / algorithmic bytes of this kernel family x the algorithmic bytes of one launch
        traffic, pmc, insitu = None, None, None
        PMC_FILE = "r05_pmc_gemv_traffic_fp8.json" if args.fp8 else "r05_pmc_gemv_traffic_bf16.json"
        try:     # rocprofv3 kernel-trace average of the same kernels inside the full decode step (committed summary of the static-batch trace)
            insitu = json.load(open(os.path.join(ROOT, "profiles", "r05_gemv_in_situ.json")))["batch32" if B > 4 else "batch1"] if (B in (1, 32) and not args.fp8) else None
        except Exception:  # noqa: BLE001
            pass
        try:     # FETCH_SIZE / WRITE_SIZE passes of this kernel family at 32 rows and at 1 (tools/gpu_lease.sh pmc_gemv; round 5: also on the fp8 stream,
                 # whose file covers gate/up + the down-projection -- the layer linears that carry 88 % of the fp8 bytes; the bf16 LM head is in the other file)
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
            traffic = round(pmc["traffic_over_algorithmic_weighted_batch32" if B > 4 else "traffic_over_algorithmic_weighted_batch1"] * bytes_per_launch)
        except Exception:  # noqa: BLE001
            pass
        traffic_source = (f"profiles/{PMC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel family): measured "
                          "traffic / algorithmic bytes ratio x the algorithmic bytes of one launch -- NOT measured in this run") if traffic else None
        roof = {"bound": "hbm", "kernel": f"k_gemv family at batch {B} (decode weight stream, all LM linears + LM head)" + (" [fp8 layer linears]" if args.fp8 else ""),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_source,
                "traffic_over_algorithmic": (pmc or {}).get("traffic_over_algorithmic_weighted_batch32" if B > 4 else "traffic_over_algorithmic_weighted_batch1"),
                "bytes_per_launch": round(bytes_per_launch), "avg_launch_us": round(avg_ms * 1e3, 2),
                "avg_launch_us_source": "HIP events around the 145-launch sequence replayed on weight-sized operands, in this run",
                "avg_launch_us_in_situ_rocprof": insitu,
                "launches_per_decode_step": n_launch,
                "decode_step_ms": round(decode_step_ms, 4) if decode_step_ms > 0 else None,
                "decode_step_achieved_GBs": round((wl + wh + kv_bytes) / (decode_step_ms * 1e-3) / 1e9, 1) if decode_step_ms > 0 else None}
        if latency is not None:
            a1 = gemv_roofline(1)
            latency["roofline"] = {"bound": "hbm", "kernel": "k_gemv family at batch 1", "achieved": round(bytes_per_launch / (a1 * 1e-3) / 1e9, 1),
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(bytes_per_launch / (a1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "avg_launch_us": round(a1 * 1e3, 2),
                                   "traffic": round(pmc["traffic_over_algorithmic_weighted_batch1"] * bytes_per_launch) if pmc else None,
                                   "decode_step_achieved_GBs": round((wl + wh + 36864.0 * (S_PROMPT + N_NEW / 2)) / (latency["decode_step_ms"] * 1e-3) / 1e9, 1)}
        del wq, wo, wg, wd, wv
        # ---- the family's HBM traffic measured in THIS run (rocprofv3 child passes, the GPU otherwise idle); without rocprofv3 the committed ratio above stays
        if world == 1 and not args.no_latency and not args.no_pmc:
            torch.cuda.empty_cache()
            live = measure_gemv_traffic(args.fp8)
            if live is not None:
                here = ("measured IN THIS RUN: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) over tools/probe_r2.py gemv on this box (FETCH_SIZE doubled per the "
                        "gfx950 note of the MI355X guide): HBM reads / algorithmic weight bytes x the algorithmic bytes of one launch")
                r32, r1 = live.get("traffic_over_algorithmic_weighted_batch32"), live.get("traffic_over_algorithmic_weighted_batch1")
                ratio = r32 if B > 4 else r1
                if ratio:
                    roof["traffic"], roof["traffic_source"], roof["traffic_over_algorithmic"] = round(ratio * bytes_per_launch), here, ratio
                if latency is not None and r1:
                    latency["roofline"]["traffic"], latency["roofline"]["traffic_over_algorithmic"] = round(r1 * bytes_per_launch), r1
        # batches of B tiles inside the timed region whose admission had the whole chip (the MFMA fractions are quoted on those)
        per = (sched["admitted"] - sched["staged_shared"]) / B if continuous else args.steps
        phases = {k: round(v / args.steps, 3) for k, v in phase_ms.items()}
        if continuous:
            phases["scheduler"] = dict({k: v // args.steps for k, v in sched.items() if k not in ("host_ms", "poll_wait_ms", "share_model")}, overlap=overlap, admit_cus_per_se=shares,
                                       share_model=sched.get("share_model"),
                                       host_ms_per_round=round(sched["host_ms"] / max(sched["rounds"], 1), 3), poll_wait_ms_per_round=round(sched["poll_wait_ms"] / max(sched["rounds"], 1), 3),
                                       decode_step_ms_shared=round(phase_ms.get("decode_shared", 0.0) / max(sched["steps_shared"], 1), 4),
                                       note=f"spans named *_shared ran concurrently on disjoint CU sets (admission share of the CUs: {args.admit_cus} of 8 per shader engine): they do not add up to ms_per_step" if overlap else None)
        vit_ms, pre_ms = phase_ms["vit"] / per, phase_ms["prefill"] / per
        phases["vit_mfma_frac"] = round(VIT_GFLOP * B / (vit_ms * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4)
        phases["prefill_mfma_frac"] = round(PREFILL_GFLOP * B / (pre_ms * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4)
        phases["forward_mfma_frac"] = round((VIT_GFLOP + PREFILL_GFLOP) * B / ((vit_ms + pre_ms) * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4)
        cpu = cpu_hf = None
        if world == 1 and not args.no_cpu_baseline:
            def dev_weight(name, shape, base):
                # synthetic parameter from the device generator (bit-identical to oracle/weights.py, pinned by
                # test_synth_fill_bit_exact) -> host float32: seconds instead of minutes of single-threaded numpy
                n = 1
                for d in shape:
                    n *= int(d)
                t = torch.empty(n, dtype=torch.bfloat16, device=dev)
                L.check(lib.sr_synth_fill(C.c_void_p(t.data_ptr()), n, name.encode(), 0, C.c_float(base), s), None, "sr_synth_fill")
                return t.float().cpu().reshape(tuple(shape))
            eng.close()                   # the GPU is idle while the host cores are timed
            cpu = cpu_baseline(dev_weight)
            try:
                cpu_hf = cpu_baseline_hf_bf16()
            except Exception as e_:  # noqa: BLE001  (transformers missing / too little host memory: reported, never fatal)
                cpu_hf = {"value": None, "error": f"{type(e_).__name__}: {e_}"[:200]}
        cfg_name = None
        if args.tile == 448 and not args.fp8:
            cfg_name = "BASELINE.json configs[2]" if B == 32 and continuous else "BASELINE.json configs[1]" if B == 1 else None
        elif args.tile == 896 and args.fp8:
            cfg_name = "BASELINE.json configs[4], one GPU's share"
        out = {
            "metric": "satellite tiles/sec (448x448, SocioReasoner-3B)" if not args.pair else "reference-faithful samples/sec (map + satellite tile per sample, SocioReasoner-3B)",
            "value": round(tiles_per_s, 4), "unit": "tiles/s" if not args.pair else "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if not args.fp8 else ("fp8-e4m3 LM linears: prefill fp8 x fp8 on the block-scaled MFMA (MX activations), decode fp8 weights x bf16 activations"
                                                      if args.fp8_mx else "bf16 (fp8-e4m3 LM linear weights, bf16 activations / MFMA)"), "data": "synthetic",
            "config": {"workload": f"SocioReasoner-3B {'fp8-weight' if args.fp8 else 'bf16'}, {B} batch row(s)/GPU, "
                                   + (f"continuous batching ({'next admission overlapped with decode' if overlap else 'admit on finish'}): {n_req} tile requests per step through {B} rows"
                                      + (" (rows drained between steps), " if args.drain else f", the {args.steps} steps served as one request stream, ") if continuous else "static batch, ")
                                   + f"{NIMG} x {args.tile}x{args.tile} synthetic image(s) per request, {S_PROMPT}-token prompt, greedy decode of {N_NEW} tokens (EOS ignored), raster tail; random-init "
                                   f"weights (counter-based generator, seed 0)" + (f" [{cfg_name}]" if cfg_name else ""),
                       "tiles_per_gpu_per_step": n_req,
                       "scheduling": ("continuous batching through B rows; the next requests' ViT + prefill are staged into spare KV slots on a CU-masked stream under the running rows' decode"
                                      if overlap else "continuous batching (admit on finish) through B rows") if continuous else "static batch",
                       "parallelism": f"dp{world}", "decode": "hipGraph" if not args.no_graph else "eager",
                       "exchange": dict(exchange, payload="float32 logits all-gather per decode step (verification mode)" if args.gather_logits
                                        else "one all-gather of 1 KB result rows per tile and step")},
            "roofline": roof, "cpu_baseline": cpu, "cpu_baseline_torch_bf16": cpu_hf, "phase_ms_per_step": phases, "static_batch": static_ref, "drained_step": drained, "more_rows_per_gpu": more_rows, "ragged": ragged, "sam2": sam, "pipeline_two_stage_with_sam2": pipeline, "latency_b1": latency,
            "weights_load_s": round(load_s, 1), "workspace_GB": round(eng.workspace_bytes / 1e9, 2),
            "result_checksum": int(res.sum().item()),
            "result_row_checksums": [int(v) for v in res.sum(dim=1).tolist()] if res.shape[0] <= 64 else None,      # one per tile of the last step, in tile order over all ranks
            "host_threads_per_rank": torch.get_num_threads(),
        }
        def clean(o):       # NaN (fractions that are not quoted for this workload) -> null
            if isinstance(o, float) and o != o:
                return None
            if isinstance(o, dict):
                return {k: clean(v) for k, v in o.items()}
            if isinstance(o, list):
                return [clean(v) for v in o]
            return o
        print(json.dumps(clean(out)), flush=True)
    dp.barrier()
    eng.close()


def measure_gemv_traffic(fp8: bool, timeout_s: int = 120):
    """HBM bytes of the decode weight-stream launches measured IN THIS RUN, on this box (hygiene: the line used to carry a
    profile-file ratio): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (kernel trace only beside them, as gpurun
    requires) over tools/probe_r2.py -- the batch-32 and batch-1 launches of this kernel family on weight-sized operands -- reduced by
    tools/rocpd_pmc.py / tools/gemv_traffic.py (FETCH_SIZE doubled per the gfx950 note of the MI355X guide).  Child processes, the GPU otherwise
    idle.  Returns the ratio dict, or None (no rocprofv3 on the box, a failed pass): the caller then falls back to the committed file."""
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TMPDIR"] = "/tmp"
    tag = "fp8" if fp8 else "bf16"
    try:
        with tempfile.TemporaryDirectory(prefix="sr_pmc_", dir="/tmp") as td:
            outs = {}
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(td, c)
                cmd = ["rocprofv3", "--kernel-trace", "--pmc", c, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", "probe_r2.py"), "gemv"] + (["fp8"] if fp8 else [])
                subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
                dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
                if not dbs:
                    return None
                outs[c] = os.path.join(td, c + ".json")
                subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_pmc.py"), dbs[0], outs[c]], capture_output=True, text=True, timeout=120, check=True)
            res = os.path.join(td, "traffic.json")
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemv_traffic.py"), outs["FETCH_SIZE"], outs["WRITE_SIZE"], res] + ([tag] if fp8 else []),
                           capture_output=True, text=True, timeout=120, check=True)
            j = json.load(open(res))
            return j if j.get("traffic_over_algorithmic_weighted_batch32") else None
    except Exception:  # noqa: BLE001  (a profiler that is missing, crashes or times out must never take the bench line with it)
        return None


def cpu_baseline(weight_source=None):
    """The oracle (a port of the reference's HF-eager CPU path) on this host's cores, on the same synthetic tile: the ViT and
    the 448-token prefill at FULL depth (32 blocks, 36 layers), then 16 greedy decode steps at full depth through the KV
    cache, extrapolated linearly to the 127 decode steps of a tile (the only extrapolation)."""
    from oracle import host_ref as H
    from oracle import model_ref as MR
    from oracle import weights as WG
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.config import geometry_3b
    cfg = MR.config_3b()
    W = WG.LazyWeights(cfg, seed=0, fast=True, source=weight_source)
    for n, _, _ in WG.param_specs(cfg):
        W[n]                                  # materialise outside the timed region
    img = synthetic.tile_pixels(0)
    ids = synthetic.tile_prompt(geometry_3b(), 0, GRID)
    p3, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], [GRID], None)
    p3 = p3[:, 0]
    nd = 16
    with torch.no_grad():
        t0 = time.perf_counter()
        pv, _ = H.patchify(img)
        emb = MR.vit_forward(W, cfg, torch.from_numpy(pv), [GRID])
        t1 = time.perf_counter()
        caches = MR.new_caches(cfg)
        x = MR.embed_with_images(W, cfg, torch.from_numpy(ids), emb)
        lg = MR.lm_forward(W, cfg, x, p3, caches)[0]
        t2 = time.perf_counter()
        base = int(p3.max()) + 1
        for k in range(nd):
            xx = W["model.embed_tokens.weight"][torch.tensor([int(lg.argmax())])]
            lg = MR.lm_forward(W, cfg, xx, torch.full((3, 1), base + k), caches)[0]
        t3 = time.perf_counter()
    vit_s, prefill_s = t1 - t0, t2 - t1
    decode_s = (N_NEW - 1) * (t3 - t2) / nd
    total = vit_s + prefill_s + decode_s
    return {"value": round(1.0 / total, 5), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/model_ref.py (float32 math, HF bf16 rounding points) on one synthetic 448x448 tile at FULL depth: 32 ViT blocks, "
                      f"36-layer 448-token prefill, {nd} greedy decode steps measured (ms/step x 127 = the tile's decode time)",
            "seconds_per_tile": round(total, 2),
            "phases_s": {"vit": round(vit_s, 2), "prefill": round(prefill_s, 2), "decode_127_steps": round(decode_s, 2),
                         "decode_s_per_step_measured": round((t3 - t2) / nd, 3)}}


def cpu_baseline_hf_bf16():
    """Library-grade CPU number beside the port: HF transformers' own Qwen2_5_VLForConditionalGeneration --
    the module the reference's hf_infer strategy calls (/root/reference/roll/distributed/strategy/hf_strategy.py:49-94) -- in bf16
    with sdpa attention on this host's cores, 3B geometry, random weights (values do not matter for the time), the same tile shape:
    ViT + 448-token prefill + 16 decode steps through the KV cache (x 127 / 16 for a tile's decode)."""
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VLForConditionalGeneration
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.config import geometry_3b
    g = geometry_3b()
    v, t = g.vision, g.text
    c = Qwen2_5_VLConfig(
        vision_config=dict(depth=v.depth, hidden_size=v.hidden_size, num_heads=v.num_heads, intermediate_size=v.intermediate_size,
                           patch_size=v.patch_size, temporal_patch_size=v.temporal_patch_size, spatial_merge_size=v.spatial_merge_size,
                           window_size=v.window_size, fullatt_block_indexes=list(v.fullatt_block_indexes), out_hidden_size=v.out_hidden_size,
                           hidden_act="silu"),
        text_config=dict(num_hidden_layers=t.num_hidden_layers, hidden_size=t.hidden_size, num_attention_heads=t.num_attention_heads,
                         num_key_value_heads=t.num_key_value_heads, intermediate_size=t.intermediate_size, vocab_size=t.vocab_size,
                         rms_norm_eps=t.rms_norm_eps, rope_parameters={"rope_type": "default", "rope_theta": t.rope_theta,
                                                                       "mrope_section": list(t.mrope_section)},
                         max_position_embeddings=32768, tie_word_embeddings=True, bos_token_id=None, eos_token_id=None),
        image_token_id=g.image_token_id, video_token_id=g.image_token_id + 1, vision_start_token_id=g.vision_start_token_id,
        vision_end_token_id=g.vision_end_token_id, tie_word_embeddings=True)
    c._attn_implementation = "sdpa"
    t0 = time.perf_counter()
    with torch.device("meta"):
        model = Qwen2_5_VLForConditionalGeneration(c)
    model = model.to(torch.bfloat16).to_empty(device="cpu").eval()
    with torch.no_grad():
        for i_, p_ in enumerate(model.parameters()):     # finite, non-zero values: CPU GEMM time does not depend on them, a random fill of 3.75 G
            p_.fill_(0.004 + 0.001 * (i_ % 7))            # elements would cost a minute of the run
        for m_ in model.modules():           # rotary inv_freq buffers were emptied with the rest
            if hasattr(m_, "inv_freq") and hasattr(m_, "original_inv_freq"):
                inv, _ = m_.compute_default_rope_parameters(m_.config)
                m_.inv_freq = inv.float()
                m_.original_inv_freq = inv.float().clone()
            elif hasattr(m_, "inv_freq") and hasattr(m_, "theta"):
                m_.inv_freq = (1.0 / (m_.theta ** (torch.arange(0, m_.dim, 2, dtype=torch.float) / m_.dim))).float()
    build_s = time.perf_counter() - t0
    from oracle import host_ref as H          # (patchify of the baseline's input only)
    pv, _ = H.patchify(synthetic.tile_pixels(0))
    pv = torch.from_numpy(pv).to(torch.bfloat16)
    ids = synthetic.tile_prompt(g, 0, GRID)
    p3, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], [GRID], None)
    S, nd = len(ids), 16
    grid_t = torch.tensor([list(GRID)])
    with torch.no_grad():
        t0 = time.perf_counter()
        o = model(input_ids=torch.from_numpy(ids)[None], attention_mask=torch.ones(1, S, dtype=torch.long), position_ids=p3,
                  pixel_values=pv, image_grid_thw=grid_t, use_cache=True)
        t1 = time.perf_counter()
        pkv, nxt, base = o.past_key_values, int(o.logits[0, -1].float().argmax()), int(p3.max()) + 1
        for k in range(nd):
            o = model(input_ids=torch.tensor([[nxt]]), attention_mask=torch.ones(1, S + k + 1, dtype=torch.long),
                      position_ids=torch.full((3, 1, 1), base + k, dtype=torch.long), past_key_values=pkv, use_cache=True)
            pkv, nxt = o.past_key_values, int(o.logits[0, -1].float().argmax())
        t2 = time.perf_counter()
    fwd_s, dec_s = t1 - t0, (N_NEW - 1) * (t2 - t1) / nd
    return {"value": round(1.0 / (fwd_s + dec_s), 5), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "torch-bf16",
            "sample": f"transformers Qwen2_5_VLForConditionalGeneration (what the reference's hf_infer strategy calls), bf16, sdpa, random weights, one synthetic "
                      f"448x448 tile at FULL depth: ViT + 448-token prefill in one forward, {nd} decode steps through the KV cache measured (x 127 / {nd})",
            "seconds_per_tile": round(fwd_s + dec_s, 2),
            "phases_s": {"vit_plus_prefill": round(fwd_s, 2), "decode_127_steps": round(dec_s, 2), "decode_s_per_step_measured": round((t2 - t1) / nd, 4)},
            "model_build_s": round(build_s, 1)}


if __name__ == "__main__":
    main()
