"""Round 5 GPU tests (through the C ABI): the fp8 weight stream above 32 rows, SAM2's float32 GEMM on the bf16 pipe, sampling without a top-k bound,
the library's switches read once.  (The tests of round 5's in-launch RMSNorms left with that code: tools/experiments/gemv_tail_head_rmsnorm.patch.)"""
import numpy as np
import pytest
import torch

from tests.util import switch

pytestmark = pytest.mark.gpu


def _prompts(rng, n, lo=5, hi=40, vocab=2000):
    ids = [rng.integers(0, vocab, int(rng.integers(lo, hi))).astype(np.int64) for _ in range(n)]
    pos = [np.tile(np.arange(len(x)), (3, 1)).astype(np.int64) for x in ids]
    return ids, pos


# ------------------------------------------------------------------------------------------------ SAM2 float32 GEMM on the bf16 matrix pipe
import ctypes as C
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sp():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _record(fname, name, payload):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, fname)
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[name] = payload
    json.dump(cur, open(path, "w"), indent=1)


@pytest.mark.parametrize("M,N,K", [(4096, 576, 144), (1024, 1152, 1152), (16384, 288, 288), (300, 144, 2304), (4096, 256, 4608)])
def test_split_bf16_gemm_is_float32_grade(monkeypatch, M, N, K):
    """sr_op_gemm_f32 computes float32 x float32 -> float32 products on the bf16 matrix pipe: each operand split exactly into three bf16 terms,
    six partial products, hi.hi in one float32 accumulator and the 2^-8-sized corrections in another (csrc/sam_f32.hip k_gemm_f32s).
    Against float64 on Hiera-L's shapes its error is that of a float32 GEMM: not larger than 1.25 x the f32-input MFMA kernel's (an exact fmaf
    chain; SR_SAM_F32_SPLIT=0) + one float32 ulp of the largest output, and the two kernels differ from each other by float32 round-off only."""
    from socioreasoner_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))          # rows of different magnitude
    W = torch.randn(N, K, generator=g) / K ** 0.5
    bias = torch.randn(N, generator=g)
    dA, dW, db = A.cuda(), W.cuda(), bias.cuda()
    want = A.double() @ W.double().t() + bias.double()
    outs, errs = {}, {}
    for flag in ("0", "1"):
        switch(monkeypatch, "SR_SAM_F32_SPLIT", flag)
        out = torch.zeros(M, N, device="cuda")
        assert L.sr_op_gemm_f32(_P(dA), K, _P(dW), M, N, K, _P(out), N, _P(db), None, None, 0, _sp()) == 0
        torch.cuda.synchronize()
        outs[flag] = out.cpu()
        errs[flag] = float((outs[flag].double() - want).abs().max())
    ulp = float(want.abs().max()) * 2.0 ** -23
    rel = float(((outs["1"] - outs["0"]).abs() / (want.abs().float() + float(want.abs().mean()))).max())
    _record("r05_split_gemm.json", f"{M}x{N}x{K}", {"max_abs_err_f32_mfma": errs["0"], "max_abs_err_split_bf16": errs["1"], "ulp_of_largest_output": ulp,
                                                   "max_rel_diff_between_kernels": rel})
    assert errs["1"] <= 1.25 * errs["0"] + ulp, errs
    assert rel < 1e-4, rel


# ------------------------------------------------------------------------------------------------ row-group GEMV (33..128 rows) at the 3B shapes, bf16 and fp8
from tests.util import assert_bf16_close, tile16x64, tile8  # noqa: E402

GV_PARTIAL, GV_SWIGLU, GV_F32, GV_BIAS, GV_RESID = range(5)
TL, XT, OT = 0x100, 0x800, 0x1000
_KEEP = []


def _D(t):
    d = t.cuda().contiguous()
    _KEEP.append(d)
    if len(_KEEP) > 48:
        torch.cuda.synchronize()
        del _KEEP[:24]
    return d


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


def _interleave16(gate, up):
    n, k = gate.shape
    out = torch.empty(2 * n, k, dtype=gate.dtype)
    o = out.view(n // 16, 2, 16, k)
    o[:, 0] = gate.view(n // 16, 16, k)
    o[:, 1] = up.view(n // 16, 16, k)
    return out


def _frag_x(x):
    """[M, K] -> the fragment-ordered activation buffer of the batch > 4 decode layer (tiled16x64 of [ceil16(M), K], zero rows appended)"""
    M, K = x.shape
    Mp = (M + 15) // 16 * 16
    xp = torch.zeros(Mp, K, dtype=x.dtype)
    xp[:M] = x
    return tile16x64(xp)


def _unfrag(t, M, K):
    Mp = (M + 15) // 16 * 16
    return t.reshape(Mp // 16, K // 64, 2, 4, 16, 8).permute(0, 4, 1, 3, 2, 5).reshape(Mp, K)[:M]


@pytest.mark.parametrize("xt", [0, 1])
@pytest.mark.parametrize("M", [33, 48, 64, 65, 100, 128])
def test_row_group_gemv_bf16_at_3b_shapes(M, xt):
    """k_gemv32g (G = 1 / 2 / 4 row groups per weight pass, row-split narrow launches) checked per operator at the 3B
    shapes against float64 / the oracle's rounding points -- q/k/v (bias), o_proj (residual, in place), gate/up (SwiGLU, row-major and
    fragment-ordered output), the 4-slab down-projection and the float32 head, with x row-major and fragment-ordered, M not a multiple
    of 16 or 32; and the refusals above 32 rows (fused norm, pending slabs)."""
    from oracle import model_ref as MR
    from socioreasoner_amd import lib
    L = lib.load()
    K, N, I = 2048, 2560, 11008
    x = _rnd((M, K), 500 + M, 1.0)
    xd = _D(_frag_x(x) if xt else x)
    fl = TL | (XT if xt else 0)
    w, b = _rnd((N, K), 501, 0.03), _rnd((N,), 502, 0.1)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    assert L.sr_op_gemv_fused(_P(xd), K, _P(_D(tile16x64(w))), M, N, K, _P(out), N, GV_BIAS | fl, _P(_D(b)), None, C.c_float(0), None, 0, None, None, None, _sp()) == 0
    assert_bf16_close(out.float().cpu(), MR.linear(x.float(), w.float(), b.float()), 1, 0.02, "qkv bias")
    wo = _rnd((2048, K), 503, 0.03)
    res = _rnd((M, 2048), 504)
    rd = res.cuda().clone()
    assert L.sr_op_gemv_fused(_P(xd), K, _P(_D(tile16x64(wo))), M, 2048, K, _P(rd), 2048, GV_RESID | fl, None, None, C.c_float(0), None, 0, None, None, None, _sp()) == 0
    assert_bf16_close(rd.float().cpu(), MR.r(res.float() + MR.linear(x.float(), wo.float())), 2, 0.01, "o_proj resid")
    g, u = _rnd((I, K), 505, 0.03), _rnd((I, K), 506, 0.03)
    wgu = _D(tile16x64(_interleave16(g, u)))
    want = MR.r(MR.silu_bf16(MR.linear(x.float(), g.float())) * MR.linear(x.float(), u.float()))
    for ot in (0, 1):          # (the fragment-ordered output needs N / 2 % 64 == 0: 11008 = 172 * 64)
        Mp = (M + 15) // 16 * 16
        act = torch.zeros(Mp * I if ot else M * I, dtype=torch.bfloat16, device="cuda")
        assert L.sr_op_gemv_fused(_P(xd), K, _P(wgu), M, 2 * I, K, _P(act), I, GV_SWIGLU | fl | (OT if ot else 0), None, None, C.c_float(0), None, 0, None, None, None, _sp()) == 0
        got = _unfrag(act.cpu(), M, I) if ot else act.cpu().reshape(M, I)
        assert_bf16_close(got.float(), want, 3, 0.005, f"gate/up swiglu out_tiled={ot}")
    xa = _rnd((M, I), 507, 0.5)
    wd = _rnd((2048, I), 508, 0.02)
    slabs = torch.zeros(4, M, 2048, dtype=torch.float32, device="cuda")
    assert L.sr_op_gemv(_P(_D(_frag_x(xa) if xt else xa)), I, _P(_D(tile16x64(wd))), M, 2048, I, _P(slabs), 4, GV_PARTIAL | fl, _sp()) == 0
    assert float((slabs.sum(0).cpu().double() - xa.double() @ wd.double().t()).abs().max()) <= 2e-3
    V = 4096
    wv = _rnd((V, K), 509, 0.03)
    nb = L.sr_op_gemv_f32_blocks(V, M, K, 0)
    lg = torch.zeros(M, V, dtype=torch.float32, device="cuda")
    av, ai = torch.zeros(M, nb, dtype=torch.float32, device="cuda"), torch.zeros(M, nb, dtype=torch.int32, device="cuda")
    assert L.sr_op_gemv_fused(_P(xd), K, _P(_D(tile16x64(wv))), M, V, K, _P(lg), V, GV_F32 | fl, None, None, C.c_float(0), None, 0, None, _P(av), _P(ai), _sp()) == 0
    torch.cuda.synchronize()
    assert float((lg.cpu().double() - x.double() @ wv.double().t()).abs().max()) <= 1e-3
    for m in (0, M - 1):
        assert int(ai[m][av[m] == av[m].max()].min()) == int(lg[m].argmax())
    # refusals above 32 rows: a fused RMSNorm prologue / pending slabs belong to the <= 4-row kernels
    nw = _D(torch.ones(K, dtype=torch.bfloat16))
    assert L.sr_op_gemv_fused(_P(xd), K, _P(_D(tile16x64(w))), M, N, K, _P(out), N, GV_BIAS | fl, _P(_D(b)), _P(nw), C.c_float(1e-6), None, 0, None, None, None, _sp()) != 0


@pytest.mark.parametrize("M", [33, 64, 100, 128])
def test_row_group_gemv_fp8_at_3b_shapes(M):
    """VERDICT round 4 (missing #3): the fp8 weight stream above 32 rows -- k_gemv32g on the tiled8 image (two 16-byte units per lane and
    64-k chunk widened exactly to the bf16 MFMA operands, per-channel scale on the float32 sums) against the oracle's quantised Linear,
    every mode the decode layer uses, x row-major and fragment-ordered."""
    from oracle import model_ref as MR
    from socioreasoner_amd import lib
    L = lib.load()
    K, N, I = 2048, 2560, 11008
    x = _rnd((M, K), 600 + M, 1.5)

    def quant(w):
        q = MR.QuantW(w.float())
        return q, _D(tile8(q.q8.view(torch.uint8))), _D(q.scale)
    for xt in (0, 1):
        xd = _D(_frag_x(x) if xt else x)
        fl = XT if xt else 0
        w, b = _rnd((N, K), 601, 0.03), _rnd((N,), 602, 0.1)
        qw, w8, sc = quant(w)
        out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        assert L.sr_op_gemv_f8(_P(xd), K, _P(w8), _P(sc), M, N, K, _P(out), N, GV_BIAS | fl, _P(_D(b)), None, C.c_float(0), 1, _sp()) == 0
        assert_bf16_close(out.float().cpu(), MR.linear(x.float(), qw, b.float()), 1, 0.02, "f8 qkv bias")
        wo = _rnd((2048, K), 603, 0.03)
        qo, o8, so = quant(wo)
        res = _rnd((M, 2048), 604)
        rd = res.cuda().clone()
        assert L.sr_op_gemv_f8(_P(xd), K, _P(o8), _P(so), M, 2048, K, _P(rd), 2048, GV_RESID | fl, None, None, C.c_float(0), 1, _sp()) == 0
        assert_bf16_close(rd.float().cpu(), MR.r(res.float() + MR.linear(x.float(), qo)), 2, 0.01, "f8 o_proj resid")
        g, u = _rnd((I, K), 605, 0.03), _rnd((I, K), 606, 0.03)
        qg, qu = MR.QuantW(g.float()), MR.QuantW(u.float())
        gu8 = _D(tile8(_interleave16(qg.q8.view(torch.uint8), qu.q8.view(torch.uint8))))
        gus = _D(_interleave16(qg.scale[:, None], qu.scale[:, None])[:, 0])
        act = torch.zeros(M, I, dtype=torch.bfloat16, device="cuda")
        assert L.sr_op_gemv_f8(_P(xd), K, _P(gu8), _P(gus), M, 2 * I, K, _P(act), I, GV_SWIGLU | fl, None, None, C.c_float(0), 1, _sp()) == 0
        want = MR.r(MR.silu_bf16(MR.linear(x.float(), qg)) * MR.linear(x.float(), qu))
        assert_bf16_close(act.float().cpu(), want, 3, 0.005, "f8 gate/up swiglu")
        xa = _rnd((M, I), 607, 0.5)
        wd = _rnd((2048, I), 608, 0.03)
        qd, d8, sd = quant(wd)
        slabs = torch.zeros(4, M, 2048, dtype=torch.float32, device="cuda")
        assert L.sr_op_gemv_f8(_P(_D(_frag_x(xa) if xt else xa)), I, _P(d8), _P(sd), M, 2048, I, _P(slabs), 2048, GV_PARTIAL | fl, None, None, C.c_float(0), 4, _sp()) == 0
        ref = (xa.double() @ qd.q.double().t()) * qd.scale.double()
        assert float((slabs.sum(0).cpu().double() - ref).abs().max()) <= 2e-3


@pytest.mark.parametrize("B,fp8", [(64, True), (128, "mx")])
def test_fp8_weights_above_32_rows_tiny_engine(B, fp8):
    """`max_batch > 32` with `lm_weight_dtype` 1 / 2 (refused until round 5: csrc/engine.hip validate): continuous batching through 64 / 128
    rows of a tiny-geometry fp8 engine with overlapped admission -- every request's tokens equal those of the same request served ALONE by
    the same engine (a row's arithmetic does not depend on its 32-row group or its neighbours), as the bf16 engine's test of round 4 asserts."""
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.engine import Engine
    from socioreasoner_amd.serving import ContinuousBatcher, Request
    geom = geometry_tiny()
    e = Engine(geom, max_patches=1024, max_prefill_tokens=64 * B, max_batch=B, max_ctx=128, max_new_tokens=24, kv_slots=2 * B, lm_fp8=fp8)
    e.load_synthetic_weights(seed=0)
    rng = np.random.default_rng(B)
    n_req = 2 * B + 5
    ids, pos = _prompts(rng, n_req)
    max_new = [int(rng.integers(3, 24)) for _ in range(n_req)]
    mk = lambda i: Request(ids=ids[i], pos3=pos[i], max_new=max_new[i], tag=i)
    cb = ContinuousBatcher(e, eos=[], pad_id=0, steps_per_poll=4, overlap=True)
    got = cb.run([mk(i) for i in range(n_req)])
    assert cb.stats["admitted"] == n_req
    for i in list(range(0, n_req, 9)) + [n_req - 1]:
        c1 = ContinuousBatcher(e, eos=[], pad_id=0, steps_per_poll=4)
        t = c1.run([mk(i)])[0]
        assert got[i] == t and len(t) == max_new[i], (i, got[i][:6], t[:6])
    e.close()


# ------------------------------------------------------------------------------------------------ multi-GPU readiness without the node
def test_bench_self_launches_8_ranks_and_every_tile_equals_the_single_process_run():
    """`python bench.py --gpus 8` with SR_DIST_BACKEND=gloo (8 ranks sharing this box's one device, host-staged exchange -- the development
    layout; on an 8-GPU node the same command runs one rank per GPU over RCCL): 8 schedulers + 8 poll loops + 8 engines on one host.  The line
    says n_gpus 8 / exchange.nranks 8 / dp8, every rank served its own tile (tile = rank), and the per-tile result rows (128 greedy tokens + the
    two IoU counts) equal, tile by tile, those of ONE process serving the same 8 tiles through one batch row.  DP contract:
    /root/reference/roll/distributed/scheduler/decorator.py:106-181 (dispatch_dp_mp_compute), protocol.py:550-617 (chunk / concat)."""
    import subprocess
    import sys
    common = ["--batch", "1", "--continuous", "--no-overlap", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-latency", "--no-sam"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SR_DIST_BACKEND")}
    r8 = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--waves", "1"] + common, cwd=ROOT, env=dict(env, SR_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=1500)
    assert r8.returncode == 0, r8.stdout[-2000:] + r8.stderr[-3000:]
    l8 = json.loads([ln for ln in r8.stdout.splitlines() if ln.startswith("{")][-1])
    assert l8["n_gpus"] == 8 and l8["config"]["exchange"]["nranks"] == 8 and l8["config"]["parallelism"] == "dp8"
    assert l8["host_threads_per_rank"] <= 8
    r1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--waves", "8"] + common, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    l1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    assert len(l8["result_row_checksums"]) == 8 and l8["result_row_checksums"] == l1["result_row_checksums"], (l8["result_row_checksums"], l1["result_row_checksums"])
    _record("r05_multi_rank.json", "8_gloo_ranks_one_device", {"tiles_per_s": l8["value"], "ms_per_step": l8["ms_per_step"], "host_threads_per_rank": l8["host_threads_per_rank"],
                                                                "scheduler": l8["phase_ms_per_step"].get("scheduler"), "single_process_tiles_per_s": l1["value"]})


# ------------------------------------------------------------------------------------------------ SAM2 float32 vs HF float32: more images, weights and prompts
def test_sam2_float32_masks_on_more_images_weights_and_multi_point_prompts(golden_dir):
    """VERDICT round 4 (weak #2): the exactness of the float32 SAM2 path was pinned on ONE image, ONE set of weights and three prompts.
    tests/golden/sam2_more.npz (tools/make_golden_sam2_more.py: HF Sam2Model float32 at Hiera-L) adds two images, a second set of synthetic
    weights and six prompts -- three clicks of mixed labels, box + three clicks, a small box in the image corner, a negative click in a box,
    four clicks, a box at the lower right edge.  Same bar as test_sam2_float32_mode_equals_hf_float32: IoU scores to 1e-4, the same best
    mask, its low-resolution logits to 1e-3, the 756 x 756 mask EXACT outside |logit| < 1e-3 and at most 10 pixels different inside."""
    from oracle import sam2_ref as S
    from socioreasoner_amd import sam2, synthetic
    g = np.load(os.path.join(golden_dir, "sam2_more.npz"))
    og = S.geometry_large()
    sg = sam2.Sam2Geometry(**{k: getattr(og, k) for k in sam2.Sam2Geometry.__dataclass_fields__})
    res, fails = {}, []
    for ci in range(int(g["n_cases"][0])):
        e = sam2.Sam2Engine(sg, dtype=torch.float32)
        e.load_state_dict(S.synthetic_weights(og, seed=int(g[f"c{ci}_weight_seed"][0])))
        e.set_image(torch.from_numpy(synthetic.tile_pixels(int(g[f"c{ci}_img_seed"][0]), 756, 756)).cuda())
        for p in range(int(g[f"c{ci}_n_prompts"][0])):
            k = f"c{ci}_p{p}"
            box = g[k + "_box"].tolist() or None
            pts = g[k + "_pts"]
            labels = g[k + "_labels"] if len(pts) else None
            logits, scores, low = e.predict(pts if len(pts) else None, labels, box, return_logits=True)
            ref_iou, best = g[k + "_iou"], int(g[k + "_best"][0])
            want = torch.from_numpy(np.unpackbits(g[k + "_mask_bits"])[: 756 * 756].reshape(756, 756).astype(bool))
            mine = torch.from_numpy(logits[best] > 0)
            d_low = float(np.abs(np.asarray(low)[best] - g[k + "_low_best"]).max())
            # HF's resized logits of the best mask (the band is defined on them): from the stored low-resolution logits, the predictor's bilinear resize
            ref3 = torch.zeros(3, *g[k + "_low_best"].shape)
            ref3[best] = torch.from_numpy(g[k + "_low_best"])
            _, _, up = S.postprocess(ref3, torch.from_numpy(ref_iou), (756, 756))
            clear = up[best].abs() >= 1e-3
            n_diff = int((mine != want).sum())
            res[k] = {"iou_max_err": float(np.abs(scores - ref_iou).max()), "low_logits_max_err_best_mask": d_low, "mask_pixels_differing_from_hf_float32": n_diff,
                      "pixels_with_abs_logit_below_1e-3": int((~clear).sum()), "mask_area": int(want.sum())}
            for ok, what in ((np.abs(scores - ref_iou).max() <= 1e-4, "iou"), (int(np.argmax(scores)) == best, "best mask"), (d_low <= 1e-3, "low-resolution logits"),
                             (bool((mine == want)[clear].all()), "mask differs outside the 1e-3 band"), (n_diff <= 10, "more than 10 pixels differ inside the band")):
                if not ok:
                    fails.append((k, what, res[k]))
        del e
        torch.cuda.empty_cache()
    _record("r05_sam2_more_parity.json", "hiera_large_float32", res)
    assert not fails, fails


# ------------------------------------------------------------------------------------------------ sampling without a top-k bound on the device (missing #5)
def _nucleus_probs(logits, temperature, top_p):
    """the sampler's rule without a top-k bound (socioreasoner_amd/sampling.py), float64"""
    x = logits.double() / temperature
    sx, si = torch.sort(x, descending=False)
    cp = torch.softmax(sx, -1).cumsum(-1)
    drop = cp <= (1.0 - top_p)
    drop[-1] = False
    sx = sx.masked_fill(drop, float("-inf"))
    return torch.softmax(torch.empty_like(x).scatter_(-1, si, sx), -1)


def test_sampling_without_top_k_bound_on_the_device():
    """vLLM's top_k = -1 (vllm_strategy.py:289-309 passes it through) used to be a host loop over sr_decode_step; k_sample_full draws on the device:
    nucleus sampling over the whole 151 936-entry vocabulary by a radix descent on probability MASS (no sort).  Checked: the support of 8192
    draws is inside the nucleus of the float64 rule and their frequencies follow it; top_p = 1 covers the tail; a tiny top_p is the arg-max; ties at the
    nucleus edge keep the lowest ids; the same (seed, row, step) gives the same token; the repetition penalty moves the choice."""
    from socioreasoner_amd import lib
    L = lib.load()
    V = 151936
    g = torch.Generator().manual_seed(9)
    base = torch.randn(V, generator=g) * 3.0

    def run(logits, B, temperature, top_p, rp=1.0, seen=None, seed=1, step=None):
        out = torch.zeros(B, dtype=torch.int64, device="cuda")
        lg = _D(logits)
        st = _D(step) if step is not None else None
        assert L.sr_op_sample(_P(lg), B, logits.shape[1], C.c_float(temperature), -1, C.c_float(top_p), C.c_float(rp), _P(_D(seen)) if seen is not None else None, seed, _P(st), _P(out),
                              None, 0, 0, _sp()) == 0
        torch.cuda.synchronize()
        return out.cpu()
    B = 8192
    rows = base[None].repeat(B, 1).contiguous()
    step = torch.arange(B, dtype=torch.int32)
    for T, tp in ((1.0, 0.8), (0.7, 0.95), (1.5, 1.0)):
        got = run(rows, B, T, tp, step=step)
        want = _nucleus_probs(base, T, tp)
        support = want > 0
        assert bool(support[got].all()), (T, tp, "a draw outside the nucleus")
        freq = torch.bincount(got, minlength=V).double() / B
        top = torch.topk(want, 40).indices
        err = (freq[top] - want[top]).abs()
        sigma = (want[top] * (1 - want[top]) / B).sqrt()
        assert bool((err <= 5 * sigma + 2e-4).all()), (T, tp, float((err / (sigma + 1e-12)).max()))
        assert torch.equal(got, run(rows, B, T, tp, step=step))                      # same (seed, row, step): same tokens
    assert bool((run(rows[:64], 64, 1.0, 1e-6, step=step[:64]) == int(base.argmax())).all())      # a nucleus of one token
    # ties at the edge of the nucleus: four equal maxima, top_p covers two and a half of them -> ids 10, 20, 30 only
    flat = torch.full((V,), -30.0)
    flat[[10, 20, 30, 40]] = 5.0
    got = run(flat[None].repeat(2048, 1).contiguous(), 2048, 1.0, 0.6, step=step[:2048])
    assert set(got.tolist()) == {10, 20, 30}, sorted(set(got.tolist()))
    # repetition penalty: the dominant token, once seen, loses its lead
    dom = torch.full((V,), -20.0)
    dom[7], dom[9] = 6.0, 5.5
    seen = torch.zeros(1, (V + 31) // 32, dtype=torch.int32)
    seen[0, 0] = 1 << 7
    a = run(dom[None].repeat(1, 1), 1, 1e-3, 0.5)
    b = run(dom[None].repeat(1, 1), 1, 1e-3, 0.5, rp=2.0, seen=seen)
    assert int(a[0]) == 7 and int(b[0]) == 9


def test_decode_sample_without_top_k_bound_tiny_engine():
    """sr_decode_sample / the rows mode accept top_k <= 0: a whole sampled decode on the device (graph-replayed step with the draw inside), same tokens eager
    and replayed, and with a vanishing top_p it is the greedy decode."""
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.engine import Engine
    geom = geometry_tiny()
    e = Engine(geom, max_patches=64, max_prefill_tokens=512, max_batch=4, max_ctx=128, max_new_tokens=16)
    e.load_synthetic_weights(seed=0)
    rng = np.random.default_rng(3)
    ids, pos = _prompts(rng, 4)
    e.prefill(ids, pos)
    greedy = e.decode(12)
    e.prefill(ids, pos)
    a = e.decode_sample(12, 1.0, -1, top_p=1e-6, seed=5, use_graph=True)
    assert torch.equal(a, greedy)
    e.prefill(ids, pos)
    s1 = e.decode_sample(12, 1.0, 0, top_p=0.9, seed=5, use_graph=True)
    e.prefill(ids, pos)
    s2 = e.decode_sample(12, 1.0, 0, top_p=0.9, seed=5, use_graph=False)
    assert torch.equal(s1, s2) and not torch.equal(s1, greedy)
    e.close()


@pytest.mark.parametrize("hw", [448, 756])
def test_prefill_attention_hand_issued_vt_reads_are_bit_identical(monkeypatch, hw):
    """k_attn_prefill2 with its V^T fragment reads issued by hand as ds_read_b64 (default) against the same kernel with the reads left to the
    compiler (SR_ATTN_VASM=0: ds_read2st64_b64 pairs, 2-way bank conflicts): same data, same MFMA order -- ViT pooler (hd 80, full-attention block),
    causal GQA prefill logits (hd 128) and the first decoded tokens are equal bit for bit.  448: four tiles, ragged prompts, partial key tiles;
    756: an image that is not a multiple of 64 patches."""
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    geom = geometry_3b()
    geom.vision.depth, geom.text.num_hidden_layers = 8, 3
    geom.vision.fullatt_block_indexes = (7,)
    tiles = {448: [0, 1, 2, 3], 756: [0]}[hw]
    n = (hw // 14) ** 2
    e = Engine(geom, max_patches=len(tiles) * n, max_prefill_tokens=len(tiles) * (n // 4 + 64), max_batch=len(tiles),
               max_ctx=(n // 4 + 64 + 63) // 64 * 64 + 64, max_new_tokens=4)
    e.load_synthetic_weights(seed=0)
    grid = (1, hw // 14, hw // 14)

    def run(vasm):
        switch(monkeypatch, "SR_ATTN_VASM", vasm)
        pix = torch.cat([e.patchify(torch.from_numpy(synthetic.tile_pixels(i, hw, hw)).cuda()) for i in tiles], dim=0)
        emb = e.vit_forward(pix, [grid] * len(tiles))
        ids, p3 = [], []
        for i in tiles:
            x = synthetic.tile_prompt(geom, i, grid, n_pre=9 + 5 * i, n_post=4 + 3 * i)
            pos3, _ = hostops.get_rope_index(torch.from_numpy(x)[None], [grid], None, image_token_id=geom.image_token_id,
                                             vision_start_token_id=geom.vision_start_token_id)
            ids.append(x)
            p3.append(pos3[:, 0].numpy())
        logits = e.prefill(ids, p3, emb, return_logits=True)
        toks = e.decode(4)
        torch.cuda.synchronize()
        return emb.clone(), logits.clone(), toks.clone()

    a, b = run("0"), run("1")
    for x, y, what in zip(a, b, ("pooler", "prefill logits", "tokens")):
        assert torch.equal(x, y), (what, float((x.float() - y.float()).abs().max()))
    assert torch.isfinite(b[1]).all()
    e.close()


def test_scheduler_measures_its_sharing_costs_and_results_do_not_depend_on_the_share():
    """ContinuousBatcher with the overlapped admission: the cost tables of the share model are MEASURED on the engine (decode slowdown / admission stretch
    per share: stats["share_model"]), kept on the engine for the next scheduler, every chosen share is a legal CU count -- and the tokens are those of
    the table-driven scheduler (SR_SCHED_ONLINE=0) and of each request decoded alone: a CU mask moves work, never a result."""
    import os
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.engine import Engine
    from socioreasoner_amd.serving import ContinuousBatcher, Request
    B = 8
    geom = geometry_tiny()
    e = Engine(geom, max_patches=1024, max_prefill_tokens=64 * B, max_batch=B, max_ctx=128, max_new_tokens=64, kv_slots=2 * B)
    e.load_synthetic_weights(seed=0)
    rng = np.random.default_rng(11)
    n_req = 6 * B
    ids = [rng.integers(0, 2000, int(rng.integers(8, 40))).astype(np.int64) for _ in range(n_req)]
    pos = [np.tile(np.arange(len(x)), (3, 1)).astype(np.int64) for x in ids]
    mk = lambda i: Request(ids=ids[i], pos3=pos[i], max_new=48, tag=i)

    def serve(online):
        old = os.environ.get("SR_SCHED_ONLINE")
        os.environ["SR_SCHED_ONLINE"] = online
        try:
            e.__dict__.pop("_sched_cal_shared", None)
            e.__dict__.pop("_sched_cal_seen", None)
            cb = ContinuousBatcher(e, eos=[], pad_id=0, steps_per_poll=4, overlap=True)
            return cb.run([mk(i) for i in range(n_req)]), cb
        finally:
            if old is None:
                os.environ.pop("SR_SCHED_ONLINE", None)
            else:
                os.environ["SR_SCHED_ONLINE"] = old

    got_off, cb_off = serve("0")
    assert not (cb_off.stats.get("share_model") or {}).get("admission_slowdown_measured"), "SR_SCHED_ONLINE=0 must not measure"
    got_on, cb_on = serve("1")
    assert got_on == got_off
    assert cb_on.stats["admitted"] == n_req and all(2 <= s <= 5 for s in cb_on.stats["shares"]), cb_on.stats["shares"]
    sm = cb_on.stats.get("share_model")
    # (a decode sample needs an admission that outlasts a whole chunk of steps: not guaranteed on this tiny geometry; the admission side always measures)
    assert sm and sm["admission_slowdown_measured"], sm
    assert all(v >= 1.0 for v in sm["decode_slowdown_measured"].values()) and all(v >= 1.0 for v in sm["admission_slowdown_measured"].values())
    c2 = ContinuousBatcher(e, eos=[], pad_id=0, steps_per_poll=4, overlap=True)      # the tables live on the engine: the next scheduler starts from them
    assert c2._adm_meas is e._sched_cal_shared[1] and c2._adm_meas
    for i in (0, n_req // 2, n_req - 1):
        assert ContinuousBatcher(e, eos=[], pad_id=0, steps_per_poll=4).run([mk(i)])[0] == got_on[i]
    e.close()
