"""Round 6 GPU tests (through the C ABI): the counted ring loops of EVERY decode GEMV launch (LDS-staged <= 4-row launches and row-major x included)
give the bits of the conditional-refill loops of rounds 1-4; configs[3] as 8 gloo ranks x 32 rows; the float32 verification path against HF float32 at 1e-3;
the x-stationary persistent gate/up and down-projection GEMVs of the CU-limited decode stream give the bits of the streaming kernels."""
from dataclasses import replace

import numpy as np
import pytest
import torch

from tests.util import switch

pytestmark = pytest.mark.gpu


def _prompts(rng, n, lo=5, hi=40, vocab=2000):
    ids = [rng.integers(0, vocab, int(rng.integers(lo, hi))).astype(np.int64) for _ in range(n)]
    pos = [np.tile(np.arange(len(x)), (3, 1)).astype(np.int64) for x in ids]
    return ids, pos


@pytest.mark.parametrize("size,fp8", [("tiny", False), ("tiny", True), ("3b", False), ("3b", True)])
def test_counted_ring_loops_equal_the_conditional_refill_loops(monkeypatch, size, fp8):
    """SR_GEMV_COUNTED=1 (default): every fill of the GEMVs' weight / x ring is unconditional (chunk index clamped to the wave's last chunk, the last
    round peeled), so hipcc emits counted vmcnt waits instead of vmcnt(0) at the top of every round.  Round 5 did this for 5..32 rows with
    fragment-ordered x; round 6 for the LDS-staged launches of <= 4 rows (RMSNorm prologue, x from LDS) and for row-major x (the down-projection at
    <= 4 rows).  Same chunks, same MFMA order: every logit of every decode step equals SR_GEMV_COUNTED=0 at 1 / 2 / 3 / 4 / 5 / 16 / 17 / 32 rows, eager and
    graph-replayed -- tiny geometry (K = 512: one ring, no steady round) and the 3B geometry 4 layers deep (K = 2048 / 11008: 2 - 4 rounds per wave)."""
    from socioreasoner_amd.config import geometry_3b, geometry_tiny
    from socioreasoner_amd.engine import Engine
    if size == "tiny":
        geom, vocab, rows = geometry_tiny(), 2000, (1, 2, 3, 4, 5, 16, 17, 32)
    else:
        geom = geometry_3b()
        geom = replace(geom, text=replace(geom.text, num_hidden_layers=4), vision=replace(geom.vision, depth=2, fullatt_block_indexes=(1,)))
        vocab, rows = 150000, (1, 3, 4, 17, 32)
    rng = np.random.default_rng(66)
    ids, pos = _prompts(rng, 32, vocab=vocab)
    out = {}
    for flag in ("0", "1"):
        switch(monkeypatch, "SR_GEMV_COUNTED", flag)
        e = Engine(geom, max_patches=64, max_prefill_tokens=64 * 32, max_batch=32, max_ctx=128, max_new_tokens=16, lm_fp8=fp8)
        e.load_synthetic_weights(seed=0)
        res = []
        for B in rows:
            e.prefill(ids[:B], pos[:B])
            toks, tr = e.decode(10, trace=True, use_graph=False)
            e.prefill(ids[:B], pos[:B])
            toks_g, tr_g = e.decode(10, trace=True, use_graph=True)
            assert torch.equal(toks, toks_g) and torch.equal(tr, tr_g), (flag, B)
            assert bool(torch.isfinite(tr).all())
            res.append((toks.clone(), tr.clone()))
        out[flag] = res
        e.close()
    for B, (t0, r0), (t1, r1) in zip(rows, out["0"], out["1"]):
        assert torch.equal(t0, t1), B
        assert torch.equal(r0, r1), (B, float((r0 - r1).abs().max()))


# ------------------------------------------------------------------------------------------------ configs[3] AS NAMED in the development layout
def test_configs3_eight_ranks_of_32_rows_equal_the_single_process_run():
    """BASELINE.json configs[3]: 8 x MI355X data-parallel, batch = 256 tiles -- 8 ranks x 32 rows.  No 8-GPU node is available to the builder, so the named
    workload runs in the development layout: `python bench.py --gpus 8 --batch 32 --waves 1` with SR_DIST_BACKEND=gloo (8 ranks = 8 engines of 10.7 GB, 8
    schedulers and 8 poll loops sharing this box's one device; host-staged exchange; on an 8-GPU node the SAME command runs one rank per GPU over RCCL and
    must print exchange.verified: true).  The line says n_gpus 8 / nranks 8 / dp8 / 256 result rows, and every tile's result row (128 greedy tokens + 2 IoU
    counts) equals, tile by tile, the row of ONE process that serves the same 256 tiles through 32 rows (8 waves, overlapped admission): sharding is
    np.array_split order, and a tile's tokens do not depend on which rank, wave or admission group served it.
    DP contract: /root/reference/roll/distributed/scheduler/decorator.py:106-181 (dispatch_dp_mp_compute), protocol.py:550-617 (chunk / concat)."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--batch", "32", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-latency", "--no-sam", "--no-pmc"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SR_DIST_BACKEND")}
    r8 = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--waves", "1"] + common, cwd=ROOT, env=dict(env, SR_DIST_BACKEND="gloo"), capture_output=True,
                        text=True, timeout=2400)
    assert r8.returncode == 0, r8.stdout[-2000:] + r8.stderr[-3000:]
    l8 = json.loads([ln for ln in r8.stdout.splitlines() if ln.startswith("{")][-1])
    assert l8["n_gpus"] == 8 and l8["config"]["exchange"]["nranks"] == 8 and l8["config"]["parallelism"] == "dp8" and l8["config"]["tiles_per_gpu_per_step"] == 32
    r1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--waves", "8"] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    l1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    assert len(l8["result_row_checksums"]) == 256 and l8["result_row_checksums"] == l1["result_row_checksums"]
    sc = l8["phase_ms_per_step"].get("scheduler") or {}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump({"workload": "BASELINE.json configs[3] in the development layout: 8 gloo ranks x 32 rows on ONE MI355X (8 engines), 256 tiles, 1 step",
               "tiles_per_s_all_ranks_one_device": l8["value"], "ms_per_step": l8["ms_per_step"], "host_threads_per_rank": l8["host_threads_per_rank"],
               "host_ms_per_round_rank0": sc.get("host_ms_per_round"), "poll_wait_ms_per_round_rank0": sc.get("poll_wait_ms_per_round"),
               "rows_equal_single_process": True, "single_process_tiles_per_s": l1["value"]}, open(os.path.join(out, "r06_configs3_gloo.json"), "w"), indent=1)


# ------------------------------------------------------------------------------------------------ north_star's literal tolerance
@pytest.mark.parametrize("tag", ["tile448", "pair448", "tile756", "tile896"])
def test_float32_path_meets_1e3_on_logits_against_hf_float32_at_full_depth(golden_dir, tag):
    """"Outputs match the reference CPU/eager path within 1e-3 on logits": with bf16 activations no implementation can (HF against itself with another
    summation order is 0.04 rms apart at this depth) -- the bf16 engine is held to HF-bf16's own distance from float32 instead
    (tests/test_gpu_round3.py).  A band of that width cannot see a SYSTEMATIC error of 1e-3, so this test closes the clause literally where it can be closed:
    the same forward with float32 activations, every FLOP through the library's float32 C-ABI entry points (tests/f32_path.py), on the same
    bf16-representable weights, at FULL depth (32 ViT blocks, 36 LM layers, vocabulary 151 936), against HF `Qwen2_5_VLForConditionalGeneration` run in float32
    (tests/golden/hf_truth3b.npz; the reference's eager caller is /root/reference/roll/distributed/strategy/hf_strategy.py:49-94):
        max |ViT + merger output - HF|, max |prefill logits - HF|, max |logits of 15 KV-cache decode steps - HF|  <=  1e-3   (|logit| up to ~4).
    What that pins to 1e-3: RMSNorm form and eps, softmax scale, rotary angles (the device cosf / sinf the engine's bf16 tables are rounded from) and the
    2-D / mRoPE channel layouts, SiLU / GELU forms (the engine's own silu_f / gelu_f), GQA mapping, causal mask, window order, merger order, KV-cache decode."""
    import json
    import os
    from oracle import host_ref as H
    from oracle import model_ref as MR
    from socioreasoner_amd import synthetic
    from tests.f32_path import F32Path
    g = np.load(os.path.join(golden_dir, "hf_truth3b.npz"))
    G, stride, ps, ls = int(g["g_new"][0]), int(g["stride"][0]), int(g["pool_stride"][0]), int(g["last_f32_stride"][0])
    cfg = MR.config_3b()
    tiles, hw = g[f"{tag}_tiles"].tolist(), int(g[f"{tag}_hw"][0])
    grid = (1, hw // 14, hw // 14)
    ids, pos3 = g[f"{tag}_ids"], g[f"{tag}_pos3"]
    f = F32Path(cfg, "cuda:0", seed=0)
    pv = torch.from_numpy(np.concatenate([H.patchify(synthetic.tile_pixels(i, hw, hw))[0] for i in tiles], axis=0)).to(torch.bfloat16).float()   # (as the fixture's runs saw them)
    emb = f.vit(pv, [grid] * len(tiles))
    res = {}
    d = (emb[:, :cfg.vision.out_hidden_size].flatten()[::ps].cpu() - torch.from_numpy(g[f"{tag}_pooler_f32"])).abs()
    res["vit_merger_max_abs_err"] = float(d.max())
    assert float(d.max()) <= 1e-3, ("ViT + merger", float(d.max()))
    logits = f.lm(f.embed(ids, emb), pos3).cpu()
    d = (logits[::ls] - torch.from_numpy(g[f"{tag}_logits_last_f32"])).abs()
    res["prefill_logits_max_abs_err"], res["logit_abs_max"] = float(d.max()), float(logits.abs().max())
    assert float(d.max()) <= 1e-3, ("prefill logits", float(d.max()))
    toks, base, worst = g[f"{tag}_tokens"].tolist(), int(pos3.max()) + 1, 0.0
    for k in range(G - 1):          # teacher-forced on the fixture's tokens, through the KV cache (hf_run in tools/make_golden_truth.py)
        lg = f.lm(f.embed([toks[k]]), np.full((3, 1), base + k)).cpu()
        e1 = float((lg[::stride] - torch.from_numpy(g[f"{tag}_sample_f32"][k])).abs().max())
        e2 = float((lg[torch.from_numpy(g[f"{tag}_top_idx"][k]).long()] - torch.from_numpy(g[f"{tag}_top_val_f32"][k])).abs().max())
        worst = max(worst, e1, e2)
        assert max(e1, e2) <= 1e-3, (k, e1, e2)
    res["decode_steps_max_abs_err"] = worst
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "r06_f32_path_vs_hf_float32.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[tag] = res
    json.dump(cur, open(path, "w"), indent=1)
    print(tag, res)


# ------------------------------------------------------------------------------------------------ x-stationary persistent GEMVs of the CU-limited decode stream
def _tile16x64(t):
    from tests.util import tile16x64
    return tile16x64(t)


@pytest.mark.parametrize("M,N,K", [(32, 22016, 2048), (17, 22016, 2048), (24, 4096, 1024), (32, 64, 512), (31, 96, 256)])
def test_x_resident_gate_up_gemv_equals_the_streaming_kernel(monkeypatch, M, N, K):
    """k_gemv_px (round 6): the 17..32-row gate/up GEMV as a persistent launch whose waves keep their K quarter of the fragment-ordered activations in registers and
    pull weight tiles from sharded ticket counters.  Per tile its arithmetic is k_gemv<SWIGLU>'s (same K quarters per wave, chunk order, MFMA order, reduction order,
    epilogue): the output equals the streaming kernel's bit for bit -- at the 3B shapes, with fewer tiles than CUs, ragged row counts, fragment-ordered and row-major
    outputs, launch after launch (the last block re-arms the ticket counters) and on a CU-masked stream (surplus blocks find no ticket)."""
    import ctypes as C
    from socioreasoner_amd import lib, streams
    L = lib.load()
    P = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
    g = torch.Generator().manual_seed(M * 7 + N + K)
    Mp = (M + 15) // 16 * 16
    x = torch.randn(Mp, K, generator=g).to(torch.bfloat16)
    x[M:] = 0
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    xt, wt = _tile16x64(x).cuda().contiguous(), _tile16x64(w).cuda().contiguous()
    outs = {}
    for flag in ("0", "1"):
        switch(monkeypatch, "SR_GEMV_XLDS", flag)
        res = []
        for out_tiled in (0x1000, 0):
            if out_tiled and (N // 2) % 64:
                continue
            for stream in (torch.cuda.current_stream(), streams.masked_stream("cuda:0", 0, 5)):
                with torch.cuda.stream(stream):
                    sp = C.c_void_p(stream.cuda_stream)
                    assert L.sr_op_gemv_set_cus(1000 if flag == "1" else 0, sp) == 0          # (n > 0 selects the x-stationary form at the op level, as sr_rows_set_cus in an engine)
                    for _ in range(3):
                        o = torch.zeros(Mp, N // 2, dtype=torch.bfloat16, device="cuda")
                        assert L.sr_op_gemv_fused(P(xt), K, P(wt), M, N, K, P(o), N // 2, 1 | 0x100 | 0x800 | out_tiled, None, None, C.c_float(0), None, 0, None,
                                                  None, None, sp) == 0
                        stream.synchronize()
                        res.append(o.cpu().clone())
        outs[flag] = res
    assert L.sr_op_gemv_set_cus(0, None) == 0
    assert len(outs["0"]) == len(outs["1"]) > 0
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b), float((a.float() - b.float()).abs().max())
    assert float(outs["1"][0].float().abs().max()) > 0


@pytest.mark.parametrize("size,fp8", [("tiny", False), ("tiny", True), ("3b", False), ("3b", True)])
def test_x_resident_gate_up_in_the_decode_step(monkeypatch, size, fp8):
    """The same switch inside the engine's captured decode step: every logit of every step at 17 / 32 rows equals SR_GEMV_XLDS=0 (bf16 and fp8 weight streams,
    eager and graph-replayed; 16 rows keep the streaming kernel either way).  The engine keeps TWO captured forms of the step -- whole chip (streaming GEMVs) and
    CU-limited stream (x-stationary GEMVs, after sr_rows_set_cus(n > 0)) -- and moves between them call by call."""
    from socioreasoner_amd.config import geometry_3b, geometry_tiny
    from socioreasoner_amd.engine import Engine
    if size == "tiny":
        geom, vocab = geometry_tiny(), 2000
    else:
        geom = geometry_3b()
        geom = replace(geom, text=replace(geom.text, num_hidden_layers=4), vision=replace(geom.vision, depth=2, fullatt_block_indexes=(1,)))
        vocab = 150000
    rng = np.random.default_rng(67)
    ids, pos = _prompts(rng, 32, vocab=vocab)
    out = {}
    for flag in ("0", "1", "3", "7", "11", "15"):
        switch(monkeypatch, "SR_GEMV_XLDS", flag)
        e = Engine(geom, max_patches=64, max_prefill_tokens=64 * 32, max_batch=32, max_ctx=128, max_new_tokens=16, lm_fp8=fp8)
        e.load_synthetic_weights(seed=0)
        res = []
        for B in (16, 17, 32):
            # the CU hint switches the step to its x-stationary form (bits 0 / 1: gate/up / down-projection; bit 2: that form on the whole chip too) and sets the deal
            # of the down-projection's tiles -- never a bit of the result
            # bit 3: the LM head with x resident in LDS (hidden size 2048 only: the "3b" cases)
            for cus in {"0": (0,), "1": (0, 160), "3": (0, 160, 37), "7": (0,), "11": (0, 160, 37), "15": (0,)}[flag]:
                e.rows_set_cus(cus)
                e.prefill(ids[:B], pos[:B])
                toks, tr = e.decode(10, trace=True, use_graph=False)
                e.prefill(ids[:B], pos[:B])
                toks_g, tr_g = e.decode(10, trace=True, use_graph=True)
                assert torch.equal(toks, toks_g) and torch.equal(tr, tr_g), (flag, B)
                if cus:
                    assert torch.equal(toks, res[-1][0]) and torch.equal(tr, res[-1][1]), (flag, B, cus)
                else:
                    res.append((toks.clone(), tr.clone()))
        out[flag] = res
        e.close()
    for flag in ("1", "3", "7", "11", "15"):
        for (t0, r0), (t1, r1) in zip(out["0"], out[flag]):
            assert torch.equal(t0, t1)
            assert torch.equal(r0, r1), float((r0 - r1).abs().max())


@pytest.mark.parametrize("M,N,K,ks", [(32, 2048, 11008, 4), (17, 2048, 11008, 4), (24, 256, 11008, 4), (32, 2048, 4096, 4), (31, 128, 7168, 4), (32, 512, 8192, 4),
                                      (32, 256, 22528, 8)])
def test_x_stationary_split_k_gemv_equals_the_streaming_kernel(monkeypatch, M, N, K, ks):
    """k_gemv32_px (round 6): the 17..32-row split-K down-projection with a block pinned to one K slab, its waves' x slice in registers and the weight tiles dealt
    statically to the first L blocks (L = sr_op_gemv_set_cus; 0 = one block per CU).  Per tile the arithmetic is k_gemv32<PARTIAL, 4>'s: the float32 slabs equal
    SR_GEMV_XLDS=0 bit for bit -- the 3B shape, ragged rows, fewer tiles than blocks, 4 / 8 / 11 chunks per wave (incl. a short last wave: 172 chunks over 16), 4 / 8
    slabs, every L (incl. L < ksplit and L > the grid), on the whole chip and on a CU-masked stream."""
    import ctypes as C
    from socioreasoner_amd import lib, streams
    L = lib.load()
    P = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
    g = torch.Generator().manual_seed(M * 7 + N + K + ks)
    x = torch.randn(32, K, generator=g).to(torch.bfloat16)
    x[M:] = 0
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    xt, wt = _tile16x64(x).cuda().contiguous(), _tile16x64(w).cuda().contiguous()
    outs = {}
    for flag in ("0", "3"):
        switch(monkeypatch, "SR_GEMV_XLDS", flag)
        res = []
        for stream in (torch.cuda.current_stream(), streams.masked_stream("cuda:0", 0, 5)):
            with torch.cuda.stream(stream):
                sp = C.c_void_p(stream.cuda_stream)
                for cus in (0, 160, 96, 7, 1, 1000):
                    assert L.sr_op_gemv_set_cus(cus, sp) == 0
                    o = torch.full((ks, M, N), float("nan"), dtype=torch.float32, device="cuda")
                    assert L.sr_op_gemv(P(xt), K, P(wt), M, N, K, P(o), ks, 0 | 0x100 | 0x800, sp) == 0
                    stream.synchronize()
                    res.append(o.cpu().clone())
                assert L.sr_op_gemv_set_cus(0, sp) == 0
        outs[flag] = res
    for a, b in zip(outs["0"], outs["3"]):
        assert torch.equal(a, b), float((a - b).abs().max())
    ref = x[:M].float() @ w.float().T
    got = outs["3"][0].sum(0)
    assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    # the fp8 weight stream (k_gemv32_px<.., F8> against k_gemv32g<PARTIAL, 4, 1, F8>)
    w8 = torch.zeros(N * K, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(N, dtype=torch.float32, device="cuda")
    assert L.sr_op_quant_f8(P(wt), N, K, P(w8), P(sc), None) == 0
    torch.cuda.synchronize()
    outs8 = {}
    for flag in ("0", "3"):
        switch(monkeypatch, "SR_GEMV_XLDS", flag)
        res = []
        for stream in (torch.cuda.current_stream(), streams.masked_stream("cuda:0", 0, 5)):
            with torch.cuda.stream(stream):
                sp = C.c_void_p(stream.cuda_stream)
                for cus in (0, 160, 7, 1000):
                    assert L.sr_op_gemv_set_cus(cus, sp) == 0
                    o = torch.full((ks, M, N), float("nan"), dtype=torch.float32, device="cuda")
                    assert L.sr_op_gemv_f8(P(xt), K, P(w8), P(sc), M, N, K, P(o), N, 0 | 0x800, None, None, C.c_float(0), ks, sp) == 0
                    stream.synchronize()
                    res.append(o.cpu().clone())
                assert L.sr_op_gemv_set_cus(0, sp) == 0
        outs8[flag] = res
    for a, b in zip(outs8["0"], outs8["3"]):
        assert torch.equal(a, b), float((a - b).abs().max())
    assert float((outs8["3"][0].sum(0) - ref).abs().max()) <= 8e-2 * float(ref.abs().max())


@pytest.mark.parametrize("M,N", [(32, 151936), (17, 151936), (25, 4096 + 96)])
def test_lm_head_with_x_in_lds_equals_the_streaming_kernel(monkeypatch, M, N):
    """k_gemv32_hpx (round 6): the 17..32-row LM head on a CU-limited stream -- x copied to LDS once per block (LDS-DMA), 4 waves walking groups of 4 vocabulary tiles,
    the weights through an 8-slot ring refilled across tile boundaries, groups dealt statically to the hinted CU count.  One wave still sums all 32 chunks of a tile in
    order, so the float32 logits AND the per-group arg-max partials equal the streaming kernel's bit for bit -- the full vocabulary, a vocabulary whose last group is
    partial (131 tiles), ragged rows, every CU hint, ordinary and CU-masked streams."""
    import ctypes as C
    from socioreasoner_amd import lib, streams
    L = lib.load()
    P = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
    K = 2048
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(32, K, generator=g).to(torch.bfloat16)
    x[M:] = 0
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    xt, wt = _tile16x64(x).cuda().contiguous(), _tile16x64(w).cuda().contiguous()
    nb = L.sr_op_gemv_f32_blocks(N, M, K, 0)
    outs = {}
    for flag in ("0", "11"):
        switch(monkeypatch, "SR_GEMV_XLDS", flag)
        res = []
        for stream in (torch.cuda.current_stream(), streams.masked_stream("cuda:0", 0, 5)):
            with torch.cuda.stream(stream):
                sp = C.c_void_p(stream.cuda_stream)
                for cus in (0, 160, 7, 1000):
                    assert L.sr_op_gemv_set_cus(cus, sp) == 0
                    o = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
                    av = torch.full((M, nb), float("nan"), dtype=torch.float32, device="cuda")
                    ai = torch.full((M, nb), -1, dtype=torch.int32, device="cuda")
                    assert L.sr_op_gemv_fused(P(xt), K, P(wt), M, N, K, P(o), N, 2 | 0x100 | 0x800, None, None, C.c_float(0), None, 0, None, P(av), P(ai), sp) == 0
                    stream.synchronize()
                    res += [o.cpu().clone(), av.cpu().clone(), ai.cpu().clone()]
                assert L.sr_op_gemv_set_cus(0, sp) == 0
        outs[flag] = res
    for a, b in zip(outs["0"], outs["11"]):
        assert torch.equal(a, b)
    ref = x[:M].float() @ w.float().T
    assert float((outs["11"][0] - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    assert torch.equal(outs["11"][2].long().gather(1, outs["11"][1].argmax(1, keepdim=True)).squeeze(1), outs["11"][0].argmax(1))


def test_streamed_and_batch_order_of_the_pipeline_write_the_same_files_on_the_real_engine(tmp_path, monkeypatch):
    """SocioSegInferPipeline.run() on the real engine (tiny synthetic weights, GREEDY decoding, so a prompt's tokens do not depend on how the scheduler grouped it):
    the streamed order -- one open request stream for both stages, 7 samples through 4 batch rows, the batch collated as 3 + 3 + 1 rows, stage-2 prompts added while
    stage-1 prompts of other samples are still waiting -- must write byte-for-byte the response texts and masks of the reference's batch order (SOCIOSEG_STREAM=0),
    and the same score."""
    import hashlib
    import os
    from roll.pipeline.rlvr.rlvr_socioseg_vlm_pipeline_infer import SocioSegInferPipeline
    from tests.test_gpu_pipeline import _cfg
    monkeypatch.setenv("SOCIOSEG_NUM_SAMPLES", "7")
    monkeypatch.setenv("SOCIOSEG_COLLATE_CHUNK", "3")
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SOCIOSEG_STREAM", mode)
        d = tmp_path / f"stream{mode}"
        cfg = _cfg(d, resp=12, prompt=1600)
        cfg["rollout_batch_size"] = 3          # rollout batches of 3 + 3 + 1 samples: the streamed order keeps ONE request stream open across them
        cfg.actor_infer.generating_args["temperature"] = 0
        pipe = SocioSegInferPipeline(cfg)
        acc = pipe.run()
        assert pipe.streamed == (mode == "1")
        res = os.path.join(str(d), "result")
        files = {}
        for sub in ("stage1", "stage2", "render1", "render2"):
            names = sorted(os.listdir(os.path.join(res, sub)))
            assert len([n for n in names if n.endswith(".png")]) == 7, (mode, sub, names)
            for n in names:
                files[f"{sub}/{n}"] = hashlib.sha256(open(os.path.join(res, sub, n), "rb").read()).hexdigest()
        out[mode] = (acc, files, [g.get("served_as") for g in pipe.actor_infer.strategy.gen_stats])
        pipe.actor_infer.strategy.engine.close()
    assert out["1"][2] == ["request stream"] and out["0"][2] == [None] * 6             # one open server against two generate calls per rollout batch
    assert out["1"][0] == out["0"][0]
    assert out["1"][1] == out["0"][1], sorted(k for k in out["1"][1] if out["1"][1][k] != out["0"][1].get(k))


def test_two_ranks_run_the_streamed_pipeline_on_their_shards(tmp_path):
    """examples/infer through torchrun with two ranks (the development layout: gloo, both ranks on this box's one device; on a node the same command is one rank
    per GPU over RCCL): every rank streams ITS shard of the samples through its own engine -- no collective inside the generation --, the per-sample IoUs are
    gathered once at the end, and both ranks report the same score.  Also pins the device index of a rank whose process group was set up by
    roll.distributed.scheduler.initialize.init before the pipeline asked dp.init_distributed (it used to keep LOCAL_RANK as its device ordinal)."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(SR_DIST_BACKEND="gloo", SCRIPTED_OBJECTS="2", OUT=str(tmp_path), SOCIOSEG_NUM_SAMPLES="20", ROLLOUT_BATCH="6", NEW_TOKENS="16", MASTER_ADDR="127.0.0.1",
               SOCIOSEG_STREAM="1")          # (forced: rollout batches of 6 through 32 rows would take the batch order on their own)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "tools", "run_example_small.py")], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 2 and all(d["streamed"] is True and d["samples"] == 20 for d in lines), lines
    assert lines[0]["giou_acc"] == lines[1]["giou_acc"]
    assert all([g.get("served_as") for g in d["generate_calls"]] == ["request stream"] for d in lines)          # one open stream per rank for its 10 samples (batches of 6 + 4)
    res = os.path.join(str(tmp_path), "result")
    for sub, n in (("stage1", 40), ("stage2", 40), ("render1", 20), ("render2", 20)):
        assert len(os.listdir(os.path.join(res, sub))) == n, (sub, len(os.listdir(os.path.join(res, sub))))
    assert open(os.path.join(res, "iou_acc.txt")).read() == f"giou_acc: {lines[0]['giou_acc']}"
