"""GPU parity tests: every call goes through the C ABI of libsocior.so and is compared with the CPU oracle.

Tolerances (written here, derived in DESIGN.md "Numerics"):
  * integer / byte / index work: bit exact;
  * a single bf16-output op on identical inputs: <= 1 bf16 ulp (of a typically sized element) on <= 1 % of the
    elements -- only float32 accumulation-order flips of a rounding are allowed;
  * float32 outputs computed from identical inputs (logits from a given hidden state): abs 1e-3;
  * composites of many rounding stages (a block, the tiny model end to end): the bf16 noise floor, bounded by
    a few ulps of the largest element -- the same bound the oracle itself meets against HF (test_oracle_golden.py).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests.util import assert_bf16_close, bf16_compare, bits_to_f32, tile16x64

pytestmark = pytest.mark.gpu

EPI_STORE, EPI_RESID, EPI_SWIGLU, EPI_GELU, EPI_F32 = range(5)
GV_PARTIAL, GV_SWIGLU, GV_F32 = range(3)


@pytest.fixture(scope="module")
def L():
    from socioreasoner_amd import lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return lib.load()


def sp():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_KEEP = []


def D(t):
    """host tensor -> device tensor that stays alive until the test module ends (the launches are asynchronous and
    a temporary freed before the next allocation could be reused by it)."""
    d = t.cuda()
    _KEEP.append(d)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return d


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


def interleave16(gate, up):
    """engine layout of gate/up rows: blocks of 16 gate rows then 16 up rows."""
    n, k = gate.shape
    out = torch.empty(2 * n, k, dtype=gate.dtype)
    o = out.view(n // 16, 2, 16, k)
    o[:, 0] = gate.view(n // 16, 16, k)
    o[:, 1] = up.view(n // 16, 16, k)
    return out


# ------------------------------------------------------------------------------------------------ generator
def test_synth_fill_bit_exact(L):
    from oracle import weights as WG
    for name, shape, base, seed in [("model.layers.3.mlp.gate_proj.weight", (1000, 333), 0.0, 0),
                                    ("visual.blocks.0.norm1.weight", (1280,), 1.0, 0),
                                    ("model.embed_tokens.weight", (4096, 257), 0.0, 5)]:
        n = int(np.prod(shape))
        out = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        assert L.sr_synth_fill(P(out), n, name.encode(), seed, C.c_float(base), sp()) == 0
        want = torch.from_numpy(WG.synth_f32(name, shape, seed, base)).reshape(-1)
        assert torch.equal(out.float().cpu(), want), name


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(1024, 1280, 1280), (448, 2560, 2048), (200, 320, 1216), (77, 512, 64), (4096, 3840, 1280)])
def test_gemm_store_bias_rowmap(L, M, N, K):
    from oracle import model_ref as MR
    a, w, b = rnd((M, K), 1), rnd((N, K), 2, 0.05), rnd((N,), 3, 0.1)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(4)).int()
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    rc = L.sr_op_gemm(P(D(a)), K, P(D(w)), M, N, K, P(out), N, P(D(b)), None, P(D(perm)), EPI_STORE, sp())
    assert rc == 0
    want = torch.empty(M, N)
    want[perm.long()] = MR.linear(a.float(), w.float(), b.float())
    assert_bf16_close(out.float().cpu(), want, 1, 0.01, f"gemm store {M}x{N}x{K}")


def test_gemm_resid_inplace_and_gelu_and_f32(L):
    from oracle import model_ref as MR
    M, N, K = 300, 1280, 3456
    a, w, b, x = rnd((M, K), 5), rnd((N, K), 6, 0.03), rnd((N,), 7, 0.1), rnd((M, N), 8)
    xd = x.cuda().clone()
    assert L.sr_op_gemm(P(D(a)), K, P(D(w)), M, N, K, P(xd), N, P(D(b)), P(xd), None, EPI_RESID, sp()) == 0
    want = MR.r(x.float() + MR.linear(a.float(), w.float(), b.float()))
    assert_bf16_close(xd.float().cpu(), want, 1, 0.01, "gemm resid")
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    assert L.sr_op_gemm(P(D(a)), K, P(D(w)), M, N, K, P(out), N, P(D(b)), None, None, EPI_GELU, sp()) == 0
    assert_bf16_close(out.float().cpu(), MR.gelu_bf16(MR.linear(a.float(), w.float(), b.float())), 2, 0.02, "gemm gelu")
    # (1 + erf) cancels for negative inputs: two float32 erf implementations differ by ~1e-4 relative there, so a
    # few tiny outputs round differently; two rounding stages -> 2 ulp)
    o32 = torch.zeros(M, N, dtype=torch.float32, device="cuda")
    assert L.sr_op_gemm(P(D(a)), K, P(D(w)), M, N, K, P(o32), N, None, None, None, EPI_F32, sp()) == 0
    ref = a.double() @ w.double().t()
    assert float((o32.cpu().double() - ref).abs().max()) <= 1e-3   # float32 accumulate vs float64


@pytest.mark.parametrize("M,I,K,bias", [(1024, 3456, 1280, True), (130, 11008, 2048, False), (64, 256, 320, True)])
def test_gemm_swiglu(L, M, I, K, bias):
    from oracle import model_ref as MR
    a, wg, wu = rnd((M, K), 9), rnd((I, K), 10, 0.03), rnd((I, K), 11, 0.03)
    bg, bu = (rnd((I,), 12, 0.1), rnd((I,), 13, 0.1)) if bias else (None, None)
    w = D(interleave16(wg, wu))
    b = D(interleave16(bg[:, None], bu[:, None]).reshape(-1)) if bias else None
    out = torch.zeros(M, I, dtype=torch.bfloat16, device="cuda")
    assert L.sr_op_gemm(P(D(a)), K, P(w), M, 2 * I, K, P(out), I, P(b), None, None, EPI_SWIGLU, sp()) == 0
    g = MR.linear(a.float(), wg.float(), bg.float() if bias else None)
    u = MR.linear(a.float(), wu.float(), bu.float() if bias else None)
    assert_bf16_close(out.float().cpu(), MR.r(MR.silu_bf16(g) * u), 3, 0.005, "gemm swiglu")   # 4 rounding stages


# ------------------------------------------------------------------------------------------------ GEMV (decode)
@pytest.mark.parametrize("M", [1, 5, 16, 17, 32])
def test_gemv_modes(L, M):
    from oracle import model_ref as MR
    K, N = 2048, 2560
    x, w = rnd((M, K), 20), rnd((N, K), 21, 0.03)
    for ks in (1, 2, 4):
        part = torch.zeros(ks, M, N, dtype=torch.float32, device="cuda")
        assert L.sr_op_gemv(P(D(x)), K, P(D(w)), M, N, K, P(part), ks, GV_PARTIAL, sp()) == 0
        got = part.sum(0).cpu()
        ref = (x.double() @ w.double().t())
        assert float((got.double() - ref).abs().max()) <= 1e-3, ("partial", M, ks)
    V = 4096
    wv = rnd((V, K), 22, 0.03)
    lg = torch.zeros(M, V, dtype=torch.float32, device="cuda")
    assert L.sr_op_gemv(P(D(x)), K, P(D(wv)), M, V, K, P(lg), 1, GV_F32, sp()) == 0
    assert float((lg.cpu().double() - x.double() @ wv.double().t()).abs().max()) <= 1e-3
    I, K2 = 11008, 2048
    wg, wu = rnd((I, K2), 23, 0.03), rnd((I, K2), 24, 0.03)
    act = torch.zeros(M, I, dtype=torch.bfloat16, device="cuda")
    assert L.sr_op_gemv(P(D(x)), K2, P(D(interleave16(wg, wu))), M, 2 * I, K2, P(act), 1, GV_SWIGLU, sp()) == 0
    want = MR.r(MR.silu_bf16(MR.linear(x.float(), wg.float())) * MR.linear(x.float(), wu.float()))
    assert_bf16_close(act.float().cpu(), want, 3, 0.005, f"gemv swiglu M={M}")
    # K = 11008 down projection, split 4
    xd, wd = rnd((M, I), 25), rnd((2048, I), 26, 0.02)
    part = torch.zeros(4, M, 2048, dtype=torch.float32, device="cuda")
    assert L.sr_op_gemv(P(D(xd)), I, P(D(wd)), M, 2048, I, P(part), 4, GV_PARTIAL, sp()) == 0
    assert float((part.sum(0).cpu().double() - xd.double() @ wd.double().t()).abs().max()) <= 2e-3


def test_tiled_weight_layout_matches_row_major(L):
    """The fragment-ordered weight layout (what the engine stores) gives bit-identical results to row-major."""
    M, N, K = 200, 2560, 2048
    a, w = rnd((M, K), 60), rnd((N, K), 61, 0.03)
    o1 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    o2 = torch.zeros_like(o1)
    assert L.sr_op_gemm(P(D(a)), K, P(D(w)), M, N, K, P(o1), N, None, None, None, EPI_STORE, sp()) == 0
    assert L.sr_op_gemm(P(D(a)), K, P(D(tile16x64(w))), M, N, K, P(o2), N, None, None, None, EPI_STORE | 0x100, sp()) == 0
    assert torch.equal(o1, o2)
    for m in (1, 20):
        x = rnd((m, K), 62)
        l1 = torch.zeros(m, N, dtype=torch.float32, device="cuda")
        l2 = torch.zeros_like(l1)
        assert L.sr_op_gemv(P(D(x)), K, P(D(w)), m, N, K, P(l1), 1, GV_F32, sp()) == 0
        assert L.sr_op_gemv(P(D(x)), K, P(D(tile16x64(w))), m, N, K, P(l2), 1, GV_F32 | 0x100, sp()) == 0
        assert torch.equal(l1, l2)
    # K = 11008 (uneven split-K), SwiGLU interleave on top of the tiling
    I = 11008
    xd, wd = rnd((3, I), 63), rnd((2048, I), 64, 0.02)
    p1 = torch.zeros(2, 3, 2048, dtype=torch.float32, device="cuda")
    p2 = torch.zeros_like(p1)
    assert L.sr_op_gemv(P(D(xd)), I, P(D(wd)), 3, 2048, I, P(p1), 2, GV_PARTIAL, sp()) == 0
    assert L.sr_op_gemv(P(D(xd)), I, P(D(tile16x64(wd))), 3, 2048, I, P(p2), 2, GV_PARTIAL | 0x100, sp()) == 0
    assert torch.equal(p1, p2)


@pytest.mark.parametrize("M", [1, 3, 4, 5, 16, 17, 32])
def test_gemv_fused_prologue_and_epilogues(L, M):
    """decode GEMV with the RMSNorm(+pending residual) prologue, bias / residual epilogues and the fused argmax."""
    from oracle import model_ref as MR
    K, N = 2048, 2560
    x, w, b = rnd((M, K), 50, 1.5), rnd((N, K), 51, 0.03), rnd((N,), 52, 0.1)
    nw = (1 + rnd((K,), 53, 0.05).float()).to(torch.bfloat16)
    slabs = torch.randn(2, M, K, generator=torch.Generator().manual_seed(54))
    h = MR.r(x.float() + MR.r(slabs[0] + slabs[1]))
    # BIAS + NORM + slabs
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    xo = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    rc = L.sr_op_gemv_fused(P(D(x)), K, P(D(w)), M, N, K, P(out), N, 3, P(D(b)), P(D(nw)), C.c_float(1e-6), P(D(slabs)), 2, P(xo),
                            None, None, sp())
    assert rc == 0
    assert_bf16_close(xo.float().cpu(), h, 1, 0.002, "pending residual write-back")
    want = MR.linear(MR.rmsnorm(h, nw.float(), 1e-6), w.float(), b.float())
    assert_bf16_close(out.float().cpu(), want, 1, 0.02, "gemv bias+norm+slabs")
    # BIAS + NORM without slabs
    assert L.sr_op_gemv_fused(P(D(x)), K, P(D(w)), M, N, K, P(out), N, 3, P(D(b)), P(D(nw)), C.c_float(1e-6), None, 0, None, None, None, sp()) == 0
    assert_bf16_close(out.float().cpu(), MR.linear(MR.rmsnorm(x.float(), nw.float(), 1e-6), w.float(), b.float()), 1, 0.02, "gemv bias+norm")
    # RESID in place (o_proj)
    res = rnd((M, 2048), 55)
    wo = rnd((2048, K), 56, 0.03)
    rd = res.cuda().clone()
    assert L.sr_op_gemv_fused(P(D(x)), K, P(D(wo)), M, 2048, K, P(rd), 2048, 4, None, None, C.c_float(0), None, 0, None, None, None, sp()) == 0
    assert_bf16_close(rd.float().cpu(), MR.r(res.float() + MR.linear(x.float(), wo.float())), 1, 0.01, "gemv resid")
    # F32 + NORM + slabs + argmax partials (vocab-sized N with a tie)
    V = 151936
    wv = rnd((V, K), 57, 0.03)
    wv[100] = wv[140000]                      # identical rows -> identical logits -> lowest index must win if it is the max
    nb = L.sr_op_gemv_f32_blocks(V, M, K, 1)
    lg = torch.zeros(M, V, dtype=torch.float32, device="cuda")
    av = torch.zeros(M, nb, dtype=torch.float32, device="cuda")
    ai = torch.zeros(M, nb, dtype=torch.int32, device="cuda")
    assert L.sr_op_gemv_fused(P(D(x)), K, P(D(wv)), M, V, K, P(lg), V, 2, None, P(D(nw)), C.c_float(1e-6), P(D(slabs)), 2, P(xo),
                              P(av), P(ai), sp()) == 0
    ref = MR.rmsnorm(h, nw.float(), 1e-6).double() @ wv.double().t()
    got = lg.cpu()
    assert float((got.double() - ref).abs().max()) <= 2e-3
    for m in range(M):
        mx = got[m].max()
        first = int((got[m] == mx).nonzero()[0, 0])
        k = int(av[m].argmax())
        cand = ai[m][av[m] == av[m].max()].min()
        assert float(av[m].max()) == float(mx) and int(cand) == first, (m, k)


# ------------------------------------------------------------------------------------------------ fp8 weight stream
@pytest.mark.parametrize("N,K", [(2560, 2048), (2048, 11008), (64, 512)])
def test_quant_f8_bit_exact(L, N, K):
    """configs[4]: the device quantiser (scale = amax/448 per output channel, q = fp8 e4m3 RNE) equals the oracle's
    definition bit for bit: scales, fp8 bytes in the tiled8 order, and the bf16 image of q left in place of W."""
    from oracle import model_ref as MR
    from tests.util import tile8
    w = rnd((N, K), 70 + N % 7, 0.03)
    w[3] = 0                                         # an all-zero row: scale 1, q 0
    w[5, 7] = 3.0                                    # an outlier sets its row's scale
    ref = MR.QuantW(w.float())
    wt = tile16x64(w).cuda().contiguous()
    w8 = torch.zeros(N * K, dtype=torch.uint8, device="cuda")
    sc = torch.zeros(N, dtype=torch.float32, device="cuda")
    assert L.sr_op_quant_f8(P(wt), N, K, P(w8), P(sc), sp()) == 0
    torch.cuda.synchronize()
    assert torch.equal(sc.cpu(), ref.scale)
    assert torch.equal(w8.cpu().view(N, K), tile8(ref.q8.view(torch.uint8)))
    assert torch.equal(wt.cpu().view(torch.int16), tile16x64(ref.q.to(torch.bfloat16)).view(torch.int16))


@pytest.mark.parametrize("M", [1, 4, 5, 17, 32])
def test_gemv_f8_modes(L, M):
    """decode GEMV on the fp8 image (widened to bf16 in registers, scale on the accumulator) against the oracle's
    quantised Linear: bias (+ fused RMSNorm at M <= 4), residual, SwiGLU, float32 K-split partials."""
    from oracle import model_ref as MR
    from tests.util import tile8
    K, N, I = 2048, 2560, 11008
    x = rnd((M, K), 80, 1.5)
    def quant(w):
        q = MR.QuantW(w.float())
        return q, tile8(q.q8.view(torch.uint8)).cuda().contiguous(), q.scale.cuda()
    # BIAS (qkv), with the norm prologue where the engine uses it
    w, b = rnd((N, K), 81, 0.03), rnd((N,), 82, 0.1)
    nw = (1 + rnd((K,), 83, 0.05).float()).to(torch.bfloat16)
    qw, w8, sc = quant(w)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    fused = M <= 4
    assert L.sr_op_gemv_f8(P(D(x)), K, P(w8), P(sc), M, N, K, P(out), N, 3, P(D(b)), P(D(nw)) if fused else None, C.c_float(1e-6), 1, sp()) == 0
    xin = MR.rmsnorm(x.float(), nw.float(), 1e-6) if fused else x.float()
    assert_bf16_close(out.float().cpu(), MR.linear(xin, qw, b.float()), 1, 0.02, "f8 gemv bias")
    # RESID (o_proj)
    wo = rnd((2048, K), 84, 0.03)
    qo, o8, so = quant(wo)
    res = rnd((M, 2048), 85)
    rd = res.cuda().clone()
    assert L.sr_op_gemv_f8(P(D(x)), K, P(o8), P(so), M, 2048, K, P(rd), 2048, 4, None, None, C.c_float(0), 1, sp()) == 0
    assert_bf16_close(rd.float().cpu(), MR.r(res.float() + MR.linear(x.float(), qo)), 1, 0.01, "f8 gemv resid")
    # SWIGLU (gate / up interleaved in blocks of 16 rows)
    g, u = rnd((I, K), 86, 0.03), rnd((I, K), 87, 0.03)
    qg, qu = MR.QuantW(g.float()), MR.QuantW(u.float())
    gu8 = tile8(interleave16(qg.q8.view(torch.uint8), qu.q8.view(torch.uint8))).cuda().contiguous()
    gus = interleave16(qg.scale[:, None], qu.scale[:, None])[:, 0].cuda().contiguous()
    act = torch.zeros(M, I, dtype=torch.bfloat16, device="cuda")
    assert L.sr_op_gemv_f8(P(D(x)), K, P(gu8), P(gus), M, 2 * I, K, P(act), I, 1, None, P(D(nw)) if fused else None, C.c_float(1e-6), 1, sp()) == 0
    want = MR.r(MR.silu_bf16(MR.linear(xin, qg)) * MR.linear(xin, qu))
    assert_bf16_close(act.float().cpu(), want, 3, 0.005, "f8 gemv swiglu")
    # PARTIAL (down-projection slabs)
    xa = rnd((M, I), 88, 0.5)
    wd = rnd((2048, I), 89, 0.03)
    qd, d8, sd = quant(wd)
    for ks in (2, 4):
        slabs = torch.zeros(ks, M, 2048, dtype=torch.float32, device="cuda")
        assert L.sr_op_gemv_f8(P(D(xa)), I, P(d8), P(sd), M, 2048, I, P(slabs), 2048, 0, None, None, C.c_float(0), ks, sp()) == 0
        ref = (xa.double() @ qd.q.double().t()) * qd.scale.double()
        assert float((slabs.sum(0).cpu().double() - ref).abs().max()) <= 2e-3, ks


# ------------------------------------------------------------------------------------------------ sampling kernel
def _filtered_probs(logits, temperature, top_k, top_p):
    """Reference distribution of the sampler (same filtering rule as socioreasoner_amd/sampling.py), float64."""
    x = logits.double() / temperature
    kth = torch.topk(x, top_k).values[-1]
    x = torch.where(x < kth, torch.full_like(x, float("-inf")), x)
    sx, si = torch.sort(x, descending=False)
    cp = torch.softmax(sx, -1).cumsum(-1)
    drop = cp <= (1.0 - top_p)
    drop[-1] = False
    sx = sx.masked_fill(drop, float("-inf"))
    return torch.softmax(torch.empty_like(x).scatter_(-1, si, sx), -1)


def test_sample_kernel_limits_and_distribution(L):
    """k_sample: top_k = 1 is the arg-max with the lowest id on ties; ties AT the top-k threshold resolve to the lowest ids;
    the repetition penalty moves the choice; and the empirical frequencies over 8192 draws follow softmax(logits / T)
    restricted by top-k and top-p (support exact, frequencies within sampling error)."""
    V = 151936
    g = torch.Generator().manual_seed(5)
    base = torch.randn(V, generator=g) * 2.0
    def run(logits, B, temperature, top_k, top_p, rp=1.0, seen=None, seed=1, step=None):
        out = torch.zeros(B, dtype=torch.int64, device="cuda")
        st = D(step) if step is not None else None
        assert L.sr_op_sample(P(D(logits)), B, logits.shape[1], C.c_float(temperature), top_k, C.c_float(top_p), C.c_float(rp),
                              P(D(seen)) if seen is not None else None, seed, P(st), P(out), None, 0, 0, sp()) == 0
        torch.cuda.synchronize()
        return out.cpu()
    # top_k = 1 (any temperature / top_p): arg-max, lowest id among equal maxima
    lg = base.repeat(3, 1).clone()
    lg[1, 777] = lg[1, 140001] = lg[1].max() + 1.0
    lg[2] = 0.0                                                   # all equal
    got = run(lg, 3, 0.9, 1, 0.5)
    assert got.tolist() == [int(base.argmax()), 777, 0]
    # ties at the threshold: rows of identical values -> the top-k set is the k lowest ids
    got = torch.stack([run(lg[2:3], 1, 1.0, 5, 1.0, seed=s_) for s_ in range(40)])
    assert set(got.flatten().tolist()) <= {0, 1, 2, 3, 4} and len(set(got.flatten().tolist())) >= 4
    # repetition penalty on the arg-max token hands the choice to the runner-up
    top2 = base.topk(2).indices.tolist()
    seen = torch.zeros(1, (V + 31) // 32, dtype=torch.int32)
    seen[0, top2[0] // 32] = 1 << (top2[0] % 32) if top2[0] % 32 < 31 else -(1 << 31)
    assert int(base[top2[0]]) >= 0 or True
    big = base.clone(); big[top2[0]] = abs(big[top2[0]]) + 1.0; big[top2[1]] = big[top2[0]] - 0.1
    assert run(big[None], 1, 1.0, 1, 1.0, rp=3.0, seen=seen).tolist() == [top2[1]]
    assert run(big[None], 1, 1.0, 1, 1.0, rp=1.0, seen=seen).tolist() == [top2[0]]
    # distribution: 32 rows x 256 steps of the same logits (the RNG is keyed by row and step): 8192 draws
    T, K, PP = 0.7, 50, 0.9
    want = _filtered_probs(base, T, K, PP)
    support = set(torch.nonzero(want > 0).flatten().tolist())
    rows = base.repeat(32, 1).cuda().contiguous()
    out = torch.zeros(32, dtype=torch.int64, device="cuda")
    out2 = torch.zeros(32, dtype=torch.int64, device="cuda")
    bmax = rows.view(32, V // 64, 64).amax(-1).contiguous()
    counts = torch.zeros(V, dtype=torch.float64)
    n_draws = 0
    for it in range(256):
        step = (torch.arange(32, dtype=torch.int32) * 256 + it).cuda()
        # odd iterations go through the block-maxima shortcut (the LM head's argmax partials): it must pick the same tokens
        if it % 2:
            assert L.sr_op_sample(P(rows), 32, V, C.c_float(T), K, C.c_float(PP), C.c_float(1.0), None, 99, P(step), P(out2), P(bmax), V // 64, 64, sp()) == 0
        assert L.sr_op_sample(P(rows), 32, V, C.c_float(T), K, C.c_float(PP), C.c_float(1.0), None, 99, P(step), P(out), None, 0, 0, sp()) == 0
        got = out.cpu()
        if it % 2:
            assert torch.equal(out2.cpu(), got), it
        assert set(got.tolist()) <= support
        counts += torch.bincount(got, minlength=V).double()
        n_draws += 32
    # chi-square-like bound: per-token |freq - p| <= 5 sigma (binomial) + 1/n
    freq = counts / n_draws
    sigma = torch.sqrt(want * (1 - want) / n_draws)
    assert bool(((freq - want).abs() <= 5 * sigma + 1.0 / n_draws).all()), float(((freq - want).abs() - 5 * sigma).max())
    assert len(support) < K and float(want.max()) < 0.9            # top-p really cut the top-k set; not degenerate


def test_decode_sample_on_device(tiny_engine, golden_dir):
    """sr_decode_sample: top_k = 1 reproduces the greedy tokens; a seed fixes the sequence (graph == eager); different
    seeds differ; every sampled token lies in the top-k set of the logits that produced it (checked by replay); eos stops."""
    g = np.load(os.path.join(golden_dir, "hf_tiny.npz"))
    grids = [tuple(x) for x in g["grids"].tolist()]
    emb = tiny_engine.vit_forward(bits_to_f32(g["pix"]).cuda(), grids)
    ids, pos3 = g["ids"], g["pos3"]
    other = ids[:9].copy()
    other[other >= 2040] = 5
    pos_o = np.tile(np.arange(9), (3, 1))
    def pre():
        return tiny_engine.prefill([ids, other], [pos3, pos_o], emb, return_logits=True)
    pre()
    greedy = tiny_engine.decode(16)
    pre()
    assert torch.equal(tiny_engine.decode_sample(16, 0.8, 1, 0.9, seed=3), greedy)
    pre()
    a = tiny_engine.decode_sample(16, 1.3, 4, 0.95, repetition_penalty=1.2, seed=7, use_graph=True)
    pre()
    b = tiny_engine.decode_sample(16, 1.3, 4, 0.95, repetition_penalty=1.2, seed=7, use_graph=False)
    pre()
    c = tiny_engine.decode_sample(16, 1.3, 4, 0.95, repetition_penalty=1.2, seed=8)
    assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, greedy)
    # replay through sr_decode_step: token i must be among the 4 best of the logits before it (penalty 1 for this check)
    pre()
    d = tiny_engine.decode_sample(12, 1.5, 4, 1.0, seed=21)
    lg = pre()
    for i in range(12):
        top = lg.topk(4, dim=-1).indices
        assert bool((d[:, i, None].long() == top).any(-1).all()), i
        lg, _ = tiny_engine.decode_step(d[:, i].long())
    # eos: the first sampled token as eos -> stop after one token, pad afterwards
    pre()
    e1 = tiny_engine.decode_sample(16, 1.5, 4, 1.0, seed=21, eos=[int(d[0, 0])], pad_id=2045)
    assert e1[0].tolist() == [int(d[0, 0])] + [2045] * 15


# ------------------------------------------------------------------------------------------------ norms / argmax
@pytest.mark.parametrize("rows,H", [(1024, 1280), (448, 2048), (3, 512), (1, 2048), (9, 320)])
def test_rmsnorm_and_resid(L, rows, H):
    from oracle import model_ref as MR
    x, w = rnd((rows, H), 30, 2.0), (1 + rnd((H,), 31, 0.05).float()).to(torch.bfloat16)
    out = torch.zeros(rows, H, dtype=torch.bfloat16, device="cuda")
    assert L.sr_op_rmsnorm(P(D(x)), P(D(w)), P(out), rows, H, C.c_float(1e-6), sp()) == 0
    assert_bf16_close(out.float().cpu(), MR.rmsnorm(x.float(), w.float(), 1e-6), 1, 0.002, "rmsnorm")
    if rows <= 32 and H <= 2048:
        part = torch.randn(3, rows, H, generator=torch.Generator().manual_seed(32))
        xd = x.cuda().clone()
        assert L.sr_op_resid_rmsnorm(P(xd), P(D(part)), 3, P(D(w)), P(out), rows, H, C.c_float(1e-6), sp()) == 0
        h = MR.r(x.float() + MR.r(part[0] + part[1] + part[2]))
        assert_bf16_close(xd.float().cpu(), h, 1, 0.002, "resid")
        assert_bf16_close(out.float().cpu(), MR.rmsnorm(h, w.float(), 1e-6), 1, 0.01, "resid norm")


def test_argmax_lowest_index_on_ties(L):
    V = 151936
    lg = torch.randn(4, V, generator=torch.Generator().manual_seed(40))
    lg[1, 77] = lg[1, 150000] = 9.0
    lg[2, V - 1] = 11.0
    lg[3, 0] = 12.0
    out = torch.zeros(4, dtype=torch.int32, device="cuda")
    assert L.sr_op_argmax(P(D(lg)), 4, V, P(out), sp()) == 0
    assert out.cpu().tolist() == [int(lg[0].argmax()), 77, V - 1, 0]


# ------------------------------------------------------------------------------------------------ raster (bit exact)
def test_raster_tail_bit_exact(L):
    from oracle import raster_ref as R
    from socioreasoner_amd import raster, synthetic
    for i in range(3):
        masks, gt = synthetic.tile_masks(i)
        masks[2] *= 255      # any non-zero byte counts
        acc = torch.zeros(756, 756, dtype=torch.uint8, device="cuda")
        for m in masks:
            raster.mask_union_(acc, torch.from_numpy(m).cuda())
        want = R.mask_union(list(masks))
        assert np.array_equal(acc.cpu().numpy(), want)
        up = raster.resize_nearest(acc, 768, 768)
        wup = R.resize_nearest(want, 768, 768)
        assert np.array_equal(up.cpu().numpy(), wup)
        assert raster.iou_counts(up, torch.from_numpy(gt).cuda()).tolist() == list(R.iou_counts(wup, gt))
        img = synthetic.tile_pixels(i)
        boxes = [[10, 20, 200, 220], [300, 5, 447, 100], [-5, -5, 30, 40], [400, 400, 500, 500], [50, 50, 40, 60]]
        got = raster.render_overlay_(torch.from_numpy(img).cuda().contiguous(), up, boxes)
        assert np.array_equal(got.cpu().numpy(), R.render_overlay(img, wup, [b for b in boxes]))
    # odd sizes, empty masks, idempotence, 896 tile
    assert raster.iou_counts(torch.zeros(64, dtype=torch.uint8, device="cuda"), torch.zeros(64, dtype=torch.uint8, device="cuda")).tolist() == [0, 0]
    assert raster.compute_giou(torch.zeros(64, dtype=torch.uint8, device="cuda"), torch.zeros(64, dtype=torch.uint8, device="cuda")) == 1.0
    big = (torch.rand(756, 756, generator=torch.Generator().manual_seed(1)) > 0.5).to(torch.uint8)
    assert np.array_equal(raster.resize_nearest(big.cuda(), 896, 896).cpu().numpy(), R.resize_nearest(big.numpy(), 896, 896))
    same = raster.resize_nearest(big.cuda(), 756, 756)
    assert torch.equal(same.cpu(), big)


@pytest.mark.parametrize("CTX,lens", [(4608, [4500, 2049, 1]), (1024, [1024, 577, 64]), (192, [191]), (640, [577, 640, 64, 1] * 4)])
def test_decode_attention_against_oracle_math(L, CTX, lens):
    """k_attn_dec_scores + k_attn_dec_pv (mRoPE of the new q/k, KV append, scores, softmax, P.V) against the oracle's
    attention arithmetic (oracle/model_ref.py lm_attention: hf:602-689 rounding points) on random caches, including the
    long-context code paths (> 1024 and > 2048 keys) that the end-to-end tests do not reach."""
    from oracle import model_ref as MR
    B, HQ, HK, HD = len(lens), 16, 2, 128
    G = HQ // HK
    torch.manual_seed(CTX + len(lens))
    qkv = (torch.randn(B, (HQ + 2 * HK) * HD) * 1.5).to(torch.bfloat16)
    kc0 = torch.randn(B, HK, CTX, HD).to(torch.bfloat16)
    vc0 = torch.randn(B, HK, HD, CTX).to(torch.bfloat16)
    ctx = torch.tensor(lens, dtype=torch.int32)
    pos = torch.tensor([min(x + 3, CTX) for x in lens], dtype=torch.int32)
    inv = 1.0 / (1e6 ** (torch.arange(0, HD, 2).float() / HD))
    ang = torch.arange(CTX + 1).float()[:, None] * inv[None]
    rc, rs = ang.cos().to(torch.bfloat16), ang.sin().to(torch.bfloat16)
    kc, vc = kc0.cuda(), vc0.cuda()
    out = torch.zeros(B, HQ * HD, dtype=torch.bfloat16, device="cuda")
    scratch = torch.zeros(B * HQ * CTX, dtype=torch.bfloat16, device="cuda")
    assert L.sr_op_attn_decode(P(D(qkv)), qkv.shape[1], P(D(pos)), P(D(ctx)), P(D(rc)), P(D(rs)), P(kc), P(vc), P(out), HQ * HD, B, HQ, HK, CTX,
                               C.c_float(HD ** -0.5), P(scratch), sp()) == 0
    torch.cuda.synchronize()
    for b, n in enumerate(lens):
        row = qkv[b].float()
        q = row[: HQ * HD].reshape(HQ, HD)
        k = row[HQ * HD: (HQ + HK) * HD].reshape(HK, HD)
        v = row[(HQ + HK) * HD:].reshape(HK, HD)
        c = torch.cat([rc[pos[b]], rc[pos[b]]]).float()
        s_ = torch.cat([rs[pos[b]], rs[pos[b]]]).float()
        q = MR.r(MR.r(q * c) + MR.r(MR.rotate_half(q) * s_))
        k = MR.r(MR.r(k * c) + MR.r(MR.rotate_half(k) * s_))
        # cache append, bit exact
        assert torch.equal(kc[b, :, n - 1].cpu().float(), k) and torch.equal(vc[b, :, :, n - 1].cpu().float(), v), b
        assert torch.equal(kc[b, :, : n - 1].cpu(), kc0[b, :, : n - 1]) and torch.equal(vc[b, :, :, : n - 1].cpu(), vc0[b, :, :, : n - 1])
        keys = torch.cat([kc0[b, :, : n - 1].float(), k[:, None]], dim=1)                   # [HK, n, HD]
        vals = torch.cat([vc0[b, :, :, : n - 1].float().transpose(1, 2), v[:, None]], dim=1)  # [HK, n, HD]
        kh, vh = keys.repeat_interleave(G, dim=0), vals.repeat_interleave(G, dim=0)
        sc = MR.r(torch.einsum("hd,hnd->hn", q, kh))
        sc = MR.r(sc * torch.tensor(HD ** -0.5))
        pr = MR.r(torch.softmax(sc, dim=-1))
        o = MR.r(torch.einsum("hn,hnd->hd", pr, vh)).reshape(-1)
        assert_bf16_close(out[b].float().cpu(), o, 2, 0.02, f"decode attention row {b} ({n} keys)")


# ------------------------------------------------------------------------------------------------ engine, tiny geometry
@pytest.fixture(scope="module")
def tiny_engine():
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.engine import Engine
    e = Engine(geometry_tiny(), max_patches=512, max_prefill_tokens=256, max_batch=4, max_ctx=192, max_new_tokens=16)
    e.load_synthetic_weights(seed=0)
    yield e
    e.close()


def test_patchify_bit_exact(tiny_engine):
    from oracle import host_ref as H
    from socioreasoner_amd import synthetic
    for (h, w) in [(56, 84), (448, 448), (112, 28)]:
        img = synthetic.tile_pixels(h * 7 + w, h, w)
        pv, grid = H.patchify(img)
        got = tiny_engine.patchify(torch.from_numpy(img).cuda())
        assert got.shape == (pv.shape[0], tiny_engine.pixel_ld)
        want = torch.from_numpy(pv).to(torch.bfloat16)
        assert torch.equal(got[:, : pv.shape[1]].cpu(), want)
        assert int(got[:, pv.shape[1]:].float().abs().sum()) == 0


def test_tiny_vit_and_prefill_and_decode(tiny_engine, golden_dir):
    """Whole tiny model against the oracle on the golden inputs (which are also pinned to HF)."""
    from oracle import model_ref as MR
    from oracle import weights as WG
    g = np.load(os.path.join(golden_dir, "hf_tiny.npz"))
    cfg = MR.config_tiny()
    W = WG.LazyWeights(cfg, seed=0)
    grids = [tuple(x) for x in g["grids"].tolist()]
    pix = bits_to_f32(g["pix"])
    img_ref = MR.vit_forward(W, cfg, pix, grids)
    img = tiny_engine.vit_forward(pix.cuda(), grids)
    mu, frac, mad = bf16_compare(img.float().cpu(), img_ref)
    assert mad <= 3 * float(img_ref.abs().max()) * 2 ** -8, ("vit", mu, frac, mad)
    # LM fed with the ORACLE's image embeddings, so that the LM comparison starts from identical inputs
    ids = g["ids"]
    pos3 = g["pos3"]
    emb = img_ref.to(torch.bfloat16).cuda()
    logits = tiny_engine.prefill([ids], [pos3], emb, return_logits=True)
    x = MR.embed_with_images(W, cfg, torch.from_numpy(ids), img_ref)
    caches = MR.new_caches(cfg)
    ref_logits = MR.lm_forward(W, cfg, x, torch.from_numpy(pos3), caches)[0]
    d = (logits[0].cpu() - ref_logits).abs()
    assert float(d.max()) <= 0.03, ("prefill logits", float(d.max()))
    # teacher-forced greedy decode: per-step logits within the noise floor, tokens equal where the margin is clear
    n_new = 12
    toks_ref, lg_ref = MR.generate_greedy(W, cfg, torch.from_numpy(ids), torch.from_numpy(pos3), img_ref, n_new)
    forced = torch.tensor([toks_ref + [0] * (16 - len(toks_ref))], dtype=torch.int32)
    tiny_engine.prefill([ids], [pos3], emb)
    toks, trace = tiny_engine.decode(16, trace=True, forced=forced, use_graph=False)
    for i in range(n_new):
        dd = (trace[i, 0].cpu() - lg_ref[i]).abs()
        assert float(dd.max()) <= 0.04, (i, float(dd.max()))
        top2 = lg_ref[i].topk(2).values
        if float(top2[0] - top2[1]) > 0.08:
            assert int(toks[0, i]) == toks_ref[i], i


def test_tiny_fp8_weights_prefill_and_decode(golden_dir):
    """configs[4] (fp8 LM linears): engine with lm_fp8 against the oracle with the same quantisation definition --
    prefill logits, teacher-forced decode logits (the decode GEMV streams the fp8 image, the prefill GEMM the bf16 image
    of the same q), graph == eager."""
    from oracle import model_ref as MR
    from oracle import weights as WG
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.engine import Engine
    g = np.load(os.path.join(golden_dir, "hf_tiny.npz"))
    cfg = MR.config_tiny()
    W = MR.Fp8LmWeights(WG.LazyWeights(cfg, seed=0))
    eng = Engine(geometry_tiny(), max_patches=512, max_prefill_tokens=256, max_batch=2, max_ctx=192, max_new_tokens=16, lm_fp8=True)
    eng.load_synthetic_weights(seed=0)
    grids = [tuple(x) for x in g["grids"].tolist()]
    img_ref = MR.vit_forward(W, cfg, bits_to_f32(g["pix"]), grids)
    ids, pos3 = g["ids"], g["pos3"]
    emb = img_ref.to(torch.bfloat16).cuda()
    logits = eng.prefill([ids], [pos3], emb, return_logits=True)
    x = MR.embed_with_images(W, cfg, torch.from_numpy(ids), img_ref)
    ref_logits = MR.lm_forward(W, cfg, x, torch.from_numpy(pos3), MR.new_caches(cfg))[0]
    d = (logits[0].cpu() - ref_logits).abs()
    assert float(d.max()) <= 0.03, ("fp8 prefill logits", float(d.max()))
    # the quantisation is visible: logits differ from the bf16 model's by more than the parity bound
    Wb = WG.LazyWeights(cfg, seed=0)
    bf = MR.lm_forward(Wb, cfg, MR.embed_with_images(Wb, cfg, torch.from_numpy(ids), img_ref), torch.from_numpy(pos3), MR.new_caches(cfg))[0]
    assert float((bf - ref_logits).abs().max()) > 0.03
    n_new = 10
    toks_ref, lg_ref = MR.generate_greedy(W, cfg, torch.from_numpy(ids), torch.from_numpy(pos3), img_ref, n_new)
    forced = torch.tensor([toks_ref + [0] * (16 - len(toks_ref))], dtype=torch.int32)
    eng.prefill([ids], [pos3], emb)
    toks, trace = eng.decode(16, trace=True, forced=forced, use_graph=False)
    for i in range(n_new):
        dd = (trace[i, 0].cpu() - lg_ref[i]).abs()
        assert float(dd.max()) <= 0.04, (i, float(dd.max()))
    eng.prefill([ids], [pos3], emb)
    a = eng.decode(16, use_graph=False)
    eng.prefill([ids], [pos3], emb)
    assert torch.equal(a, eng.decode(16, use_graph=True))
    eng.close()


def test_vit_random_grids_window_index_math(tiny_engine):
    """The engine's host-side ViT index math (window permutation, cu_seqlens of windows / images, 2-D rotary tables, merger
    un-permutation -- engine.hip vit_prepare) on seeded random ragged grids, 1-3 images per call, against the oracle
    (which is pinned to HF's get_vision_window_index on the golden grids)."""
    from oracle import model_ref as MR
    from oracle import weights as WG
    cfg = MR.config_tiny()
    W = WG.LazyWeights(cfg, seed=0)
    rng = np.random.default_rng(21)
    for case in range(10):
        grids, n = [], 0
        for _ in range(int(rng.integers(1, 4))):
            h, w = 2 * int(rng.integers(1, 12)), 2 * int(rng.integers(1, 12))
            if n + h * w > 480:
                continue
            grids.append((1, h, w))
            n += h * w
        if not grids:
            grids, n = [(1, 6, 10)], 60
        pix = torch.from_numpy(rng.standard_normal((n, 1176)).astype(np.float32)).to(torch.bfloat16).float()
        ref = MR.vit_forward(W, cfg, pix, grids)
        got = tiny_engine.vit_forward(pix.cuda(), grids)
        assert got.shape == ref.shape, (grids, got.shape, ref.shape)
        mu, frac, mad = bf16_compare(got.float().cpu(), ref)
        assert mad <= 3 * float(ref.abs().max()) * 2 ** -8, (case, grids, mu, frac, mad)


def test_decode_graph_equals_eager_and_batch_invariance(tiny_engine, golden_dir):
    g = np.load(os.path.join(golden_dir, "hf_tiny.npz"))
    grids = [tuple(x) for x in g["grids"].tolist()]
    emb = tiny_engine.vit_forward(bits_to_f32(g["pix"]).cuda(), grids)
    ids, pos3 = g["ids"], g["pos3"]
    tiny_engine.prefill([ids], [pos3], emb)
    eager = tiny_engine.decode(16, use_graph=False)
    tiny_engine.prefill([ids], [pos3], emb)
    graph = tiny_engine.decode(16, use_graph=True)
    assert torch.equal(eager, graph)
    # the same sequence in three slots of a batch, plus a shorter text-only neighbour: identical tokens
    other = ids[:9].copy()
    other[other >= 2040] = 5
    pos_o = np.tile(np.arange(9), (3, 1))
    emb3 = torch.cat([emb, emb, emb])
    tiny_engine.prefill([ids, other, ids, ids], [pos3, pos_o, pos3, pos3], emb3)
    bt = tiny_engine.decode(16, use_graph=True)
    assert torch.equal(bt[0], eager[0]) and torch.equal(bt[2], eager[0]) and torch.equal(bt[3], eager[0])
    # eos handling: stop at the first generated token, pad afterwards
    first = int(eager[0, 0])
    tiny_engine.prefill([ids], [pos3], emb)
    st = tiny_engine.decode(16, eos=[first], pad_id=2045, use_graph=False)
    assert st[0].tolist() == [first] + [2045] * 15


def test_decode_step_matches_decode_and_accepts_caller_tokens(tiny_engine, golden_dir):
    """sr_decode_step: greedy chaining reproduces sr_decode bit for bit (tokens and logits); a token
    chosen by the caller takes the place of the argmax exactly as teacher forcing does; context overflow is refused."""
    g = np.load(os.path.join(golden_dir, "hf_tiny.npz"))
    grids = [tuple(x) for x in g["grids"].tolist()]
    emb = tiny_engine.vit_forward(bits_to_f32(g["pix"]).cuda(), grids)
    ids, pos3 = g["ids"], g["pos3"]
    other = ids[:9].copy()
    other[other >= 2040] = 5
    pos_o = np.tile(np.arange(9), (3, 1))
    tiny_engine.prefill([ids, other], [pos3, pos_o], emb)
    toks, trace = tiny_engine.decode(16, trace=True, use_graph=False)
    lg0 = tiny_engine.prefill([ids, other], [pos3, pos_o], emb, return_logits=True)
    assert torch.equal(lg0, trace[0])
    got = [lg0.argmax(-1)]
    lgs = [lg0]
    for i in range(6):
        lg, nxt = tiny_engine.decode_step(None if i % 2 else got[-1])      # engine's own greedy token / caller's (same) token
        assert torch.equal(nxt, lg.argmax(-1))
        got.append(nxt)
        lgs.append(lg)
    assert torch.equal(torch.stack(got, 1).int(), toks[:, :7])
    for i in range(7):
        assert torch.equal(lgs[i], trace[i]), i
    # caller-chosen tokens == teacher forcing
    forced = torch.randint(0, 2000, (2, 16), dtype=torch.int32)
    tiny_engine.prefill([ids, other], [pos3, pos_o], emb)
    _, tr_f = tiny_engine.decode(16, trace=True, forced=forced, use_graph=False)
    tiny_engine.prefill([ids, other], [pos3, pos_o], emb)
    for i in range(5):
        lg, _ = tiny_engine.decode_step(forced[:, i].long().cuda())
        assert torch.equal(lg, tr_f[i + 1]), i
    # capacity: max_ctx 192, the long prompt has len(ids) tokens
    with pytest.raises(Exception, match="max_ctx"):
        for _ in range(200):
            tiny_engine.decode_step(None, return_logits=False)


def test_continuous_batching_matches_one_at_a_time(tiny_engine, golden_dir):
    """configs[2]: requests admitted into free rows while others are mid-decode (admit on finish), each with its own length
    limit / eos, produce exactly the tokens of the same request served alone."""
    from socioreasoner_amd.serving import ContinuousBatcher, Request
    g = np.load(os.path.join(golden_dir, "hf_tiny.npz"))
    grids = [tuple(x) for x in g["grids"].tolist()]
    pix = bits_to_f32(g["pix"]).cuda()
    ids_img, pos_img = g["ids"], g["pos3"]
    emb = tiny_engine.vit_forward(pix, grids)
    rng = np.random.default_rng(3)
    reqs = []
    plan = [(9, 16), (30, 5), (17, 11), (None, 16), (4, 7), (25, 16), (None, 3), (12, 9), (40, 1)]
    plan += [(int(rng.integers(1, 60)) if rng.random() > 0.15 else None, int(rng.integers(1, 17))) for _ in range(24)]   # churn
    for k, (n, mx) in enumerate(plan):
        if n is None:
            reqs.append((ids_img, pos_img, mx, True))
        else:
            ids = rng.integers(0, 2000, n).astype(np.int64)
            reqs.append((ids, np.tile(np.arange(n), (3, 1)), mx, False))
    # reference: every request alone through the static path
    alone = []
    for ids, pos, mx, has_img in reqs:
        tiny_engine.prefill([ids], [pos], emb if has_img else None)
        alone.append(tiny_engine.decode(16, use_graph=False)[0, :mx].tolist())
    eos = [alone[0][6], alone[5][9]]                       # tokens that occur mid-sequence somewhere -> early stops

    def cut(row):
        for j, t in enumerate(row):
            if t in eos:
                return row[: j + 1]
        return row
    want = [cut(a) for a in alone]
    assert any(len(w) < len(a) for w, a in zip(want, alone))

    class PixEngine:                                        # the batcher patchifies uint8 images; here features are ready-made
        def __init__(self, e):
            self.e = e
        def __getattr__(self, k):
            return getattr(self.e, k)
        def patchify(self, im):
            return im
        def vit_forward(self, pixrows, gr):
            return self.e.vit_forward(pixrows, gr)
    for steps_per_poll in (1, 4):
        cb = ContinuousBatcher(PixEngine(tiny_engine), eos, pad_id=2045, steps_per_poll=steps_per_poll)
        rl = [Request(ids=i, pos3=p, max_new=mx, images=[pix] if has_img else [], grids=grids if has_img else []) for i, p, mx, has_img in reqs]
        got = cb.run(rl)
        assert got == want, (steps_per_poll, got, want)
        assert cb.stats["admissions"] >= 8                  # 33 requests through 4 rows: rows were re-used mid-flight
    # sampling rows: top_k = 1 is the greedy result again; a seed fixes the outcome, another seed changes it
    def run_sampled(**smp):
        cb = ContinuousBatcher(PixEngine(tiny_engine), eos, pad_id=2045, steps_per_poll=2, sampling=smp)
        return cb.run([Request(ids=i, pos3=p, max_new=mx, images=[pix] if has_img else [], grids=grids if has_img else []) for i, p, mx, has_img in reqs])
    assert run_sampled(temperature=0.7, top_k=1, top_p=0.9, seed=1) == want
    s1 = run_sampled(temperature=1.4, top_k=6, top_p=0.95, seed=5)
    assert s1 == run_sampled(temperature=1.4, top_k=6, top_p=0.95, seed=5) and s1 != run_sampled(temperature=1.4, top_k=6, top_p=0.95, seed=6)
    assert s1 != want and all(1 <= len(t) <= mx for t, (_, _, mx, _) in zip(s1, reqs))
    # static calls work again afterwards
    tiny_engine.prefill([reqs[0][0]], [reqs[0][1]], None)
    assert tiny_engine.decode(16, use_graph=True)[0].tolist() == alone[0]
    with pytest.raises(Exception, match="rows"):
        tiny_engine.rows_step(1)


# ------------------------------------------------------------------------------------------------ true dimensions
def _truedim(full: bool):
    from oracle import model_ref as MR
    from socioreasoner_amd.config import geometry_3b
    geom, cfg = geometry_3b(), MR.config_3b()
    for c in (geom, cfg):
        c.vision.depth, c.text.num_hidden_layers, c.text.vocab_size = 1, 1, 4096
        c.vision.fullatt_block_indexes = (0,) if full else ()
        c.image_token_id, c.vision_start_token_id, c.vision_end_token_id = 4000, 4001, 4002
    return geom, cfg


def test_truedim_single_block_and_layer():
    """3B dimensions, one ViT block (window / full attention) + merger, one LM layer + LM head slice, one decode
    step through the KV cache: engine vs oracle."""
    from oracle import host_ref as H
    from oracle import model_ref as MR
    from oracle import weights as WG
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.engine import Engine
    img = synthetic.tile_pixels(0)
    pv, grid = H.patchify(img)
    ref_img = None
    for full in (False, True):
        geom, cfg = _truedim(full)
        W = WG.LazyWeights(cfg, seed=0)
        e = Engine(geom, max_patches=1024, max_prefill_tokens=512, max_batch=2, max_ctx=576, max_new_tokens=8)
        e.load_synthetic_weights(seed=0)
        ref = MR.vit_forward(W, cfg, torch.from_numpy(pv), [grid])
        got = e.vit_forward(e.patchify(torch.from_numpy(img).cuda()), [grid]) if not full else \
            e.vit_forward(torch.from_numpy(pv).cuda(), [grid])                    # float32 pixel input path
        mu, frac, mad = bf16_compare(got.float().cpu(), ref)
        assert mad <= 2 * float(ref.abs().max()) * 2 ** -8, ("vit block full=%s" % full, mu, frac, mad)
        if full:
            e.close()
            continue
        ref_img = ref
        ids = synthetic.tile_prompt(geom, 0, grid)
        assert len(ids) == 448
        pos3, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], [grid], None, image_token_id=4000, vision_start_token_id=4001)
        pos3 = pos3[:, 0].numpy()
        logits = e.prefill([ids], [pos3], ref_img.to(torch.bfloat16).cuda(), return_logits=True)
        x = MR.embed_with_images(W, cfg, torch.from_numpy(ids), ref_img)
        caches = MR.new_caches(cfg)
        ref_logits = MR.lm_forward(W, cfg, x, torch.from_numpy(pos3), caches)[0]
        d = (logits[0].cpu() - ref_logits).abs()
        assert float(d.max()) <= 0.02, ("1-layer logits", float(d.max()), float(ref_logits.abs().max()))
        tok = int(ref_logits.argmax())
        forced = torch.full((1, 8), tok, dtype=torch.int32)
        _, trace = e.decode(8, trace=True, forced=forced, use_graph=False)
        xx = W["model.embed_tokens.weight"][torch.tensor([tok])]
        p3 = torch.full((3, 1), int(pos3.max()) + 1, dtype=torch.int64)
        ref_step = MR.lm_forward(W, cfg, xx, p3, caches)[0]
        d = (trace[1, 0].cpu() - ref_step).abs()
        assert float(d.max()) <= 0.02, ("decode step logits", float(d.max()))
        e.close()


def test_truedim_ragged_windows_756_and_896():
    """BASELINE.json configs[4] geometry: an 896x896 tile is 756x756 under the reference's default max_pixels (ragged
    36..64-patch windows, 2916 patches, not a multiple of 64) and 896x896 when max_pixels is honoured (4096 patches).
    One ViT block (window, then full attention) + merger at the true dimensions against the oracle."""
    from oracle import host_ref as H
    from oracle import model_ref as MR
    from oracle import weights as WG
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.engine import Engine
    assert hostops.smart_resize(896, 896) == (756, 756) and hostops.smart_resize(896, 896, max_pixels=1344 * 1344) == (896, 896)
    for side, full in ((756, False), (756, True), (896, False)):
        geom, cfg = _truedim(full)
        W = WG.LazyWeights(cfg, seed=0)
        e = Engine(geom, max_patches=4096, max_prefill_tokens=64, max_batch=1, max_ctx=64, max_new_tokens=4)
        e.load_synthetic_weights(seed=0)
        img = synthetic.tile_pixels(7, side, side)
        pv, grid = H.patchify(img)
        assert grid == (1, side // 14, side // 14)
        ref = MR.vit_forward(W, cfg, torch.from_numpy(pv), [grid])
        got = e.vit_forward(e.patchify(torch.from_numpy(img).cuda()), [grid])
        mu, frac, mad = bf16_compare(got.float().cpu(), ref)
        assert mad <= 2 * float(ref.abs().max()) * 2 ** -8, (side, full, mu, frac, mad)
        e.close()


# ------------------------------------------------------------------------------------------------ full size (3B), properties
def test_full_size_3b_properties():
    """BASELINE.json's full geometry (SocioReasoner-3B, 448x448 tile, 448-token prompt): size-independent properties.
    (The CPU oracle needs minutes for 36+32 layers; layer-level parity at these dimensions is test_truedim_*.)
      * decode through the captured hipGraph == eager launches, bit for bit, and run-to-run deterministic;
      * batch invariance: the same tile in two slots of a batch (next to a different tile) decodes to the same tokens;
      * KV-cache consistency: the logits of decode step k equal (within the bf16 noise floor) the prefill logits of
        the prompt extended by the k forced tokens -- two different kernel paths (GEMV + decode attention vs GEMM +
        prefill attention) computing the same function."""
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    geom = geometry_3b()
    e = Engine(geom, max_patches=3072, max_prefill_tokens=1536, max_batch=3, max_ctx=640, max_new_tokens=16)
    e.load_synthetic_weights(seed=0)
    grid = (1, 32, 32)

    def prep(i, extra=()):
        ids = np.concatenate([synthetic.tile_prompt(geom, i, grid), np.asarray(extra, dtype=np.int64)])
        p, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], [grid], None)
        return ids, p[:, 0].numpy()
    pix = [e.patchify(torch.from_numpy(synthetic.tile_pixels(i)).cuda()) for i in (0, 1)]
    emb0 = e.vit_forward(pix[0], [grid])
    emb01 = e.vit_forward(torch.cat([pix[0], pix[1], pix[0]]), [grid] * 3)
    assert torch.equal(emb01[:256], emb0) and torch.equal(emb01[512:], emb0)      # ViT batch invariance, bit exact
    ids0, pos0 = prep(0)
    ids1, pos1 = prep(1)
    e.prefill([ids0], [pos0], emb0)
    eager, trace = e.decode(16, trace=True, use_graph=False)
    e.prefill([ids0], [pos0], emb0)
    graph = e.decode(16, use_graph=True)
    e.prefill([ids0], [pos0], emb0)
    graph2 = e.decode(16, use_graph=True)
    assert torch.equal(eager, graph) and torch.equal(graph, graph2)
    e.prefill([ids0, ids1, ids0], [pos0, pos1, pos0], emb01)
    bt = e.decode(16, use_graph=True)
    assert torch.equal(bt[0], eager[0]) and torch.equal(bt[2], eager[0]) and not torch.equal(bt[1], eager[0])
    # all-position logits (sr_forward_logits: final norm over every row + LM head as an MFMA GEMM, N = 151 936) against the
    # last-position logits of the prefill (LM head as the decode GEMV): same function, float32 accumulation order apart
    allp = e.forward_logits([ids0], [pos0], emb0)
    last = e.prefill([ids0], [pos0], emb0, return_logits=True)
    assert allp.shape == (len(ids0), geom.text.vocab_size) and bool(torch.isfinite(allp).all())
    assert float((allp[-1] - last[0]).abs().max()) <= 2e-3
    toks = eager[0].tolist()
    for k in (1, 5):
        idk, posk = prep(0, toks[:k])
        lg = e.prefill([idk], [posk], emb0, return_logits=True)
        d = (lg[0] - trace[k, 0]).abs()
        scale = float(trace[k, 0].abs().max())
        # 36 layers x ~12 bf16 rounding stages: the noise floor of two correct implementations is a few % of |logit|max
        assert float(d.max()) <= 0.08 * max(scale, 1.0), (k, float(d.max()), scale)
        top2 = trace[k, 0].topk(2).values
        if float(top2[0] - top2[1]) > 0.16 * scale:
            assert int(lg[0].argmax()) == int(trace[k, 0].argmax())
    e.close()


def test_full_size_config5_fp8_896_tile():
    """BASELINE.json configs[4] at full size on one GPU: fp8 LM linears, one 896 x 896 tile (4096 patches -> 1024 image
    tokens, full-attention blocks over 4096 keys, S = 1216 prompt).  Size-independent properties: graph == eager and
    deterministic; ViT batch invariance; KV-cache consistency between the fp8 decode GEMV path and the prefill GEMM path
    (bf16 image of the same quantised weights) within the bf16 noise floor."""
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    geom = geometry_3b()
    e = Engine(geom, max_patches=8192, max_prefill_tokens=1280, max_batch=2, max_ctx=1280, max_new_tokens=8, lm_fp8=True)
    e.load_synthetic_weights(seed=0)
    grid = (1, 64, 64)

    def prep(i, extra=()):
        ids = np.concatenate([synthetic.tile_prompt(geom, i, grid), np.asarray(extra, dtype=np.int64)])
        p, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], [grid], None)
        return ids, p[:, 0].numpy()
    pix0 = e.patchify(torch.from_numpy(synthetic.tile_pixels(0, 896, 896)).cuda())
    pix1 = e.patchify(torch.from_numpy(synthetic.tile_pixels(1, 896, 896)).cuda())
    assert pix0.shape[0] == 4096
    emb0 = e.vit_forward(pix0, [grid])
    emb10 = e.vit_forward(torch.cat([pix1, pix0]), [grid] * 2)
    assert emb0.shape == (1024, 2048) and torch.equal(emb10[1024:], emb0) and bool(torch.isfinite(emb0.float()).all())
    ids0, pos0 = prep(0)
    assert len(ids0) == 1216
    e.prefill([ids0], [pos0], emb0)
    eager, trace = e.decode(8, trace=True, use_graph=False)
    e.prefill([ids0], [pos0], emb0)
    graph = e.decode(8, use_graph=True)
    assert torch.equal(eager, graph) and bool(torch.isfinite(trace).all())
    toks = eager[0].tolist()
    idk, posk = prep(0, toks[:3])
    lg = e.prefill([idk], [posk], emb0, return_logits=True)
    d = (lg[0] - trace[3, 0]).abs()
    scale = float(trace[3, 0].abs().max())
    assert float(d.max()) <= 0.08 * max(scale, 1.0), (float(d.max()), scale)
    e.close()


def test_full_size_batch_variants_agree():
    """Full 3B geometry across the batch-size dispatch boundaries of the decode path (<= 4: fused norm prologues; 5..16:
    16-row tiles; 17..32: 32-row LM head / 4-slab down-projection): within one kernel family the tokens of a tile do not
    depend on the batch around it (5 vs 16, 17 vs 32), every family is deterministic and graph == eager, and the families
    agree with each other on the first token (same prefill path) and stay within the noise floor on later logits."""
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    geom = geometry_3b()
    e = Engine(geom, max_patches=1024, max_prefill_tokens=448 * 32, max_batch=32, max_ctx=512, max_new_tokens=8)
    e.load_synthetic_weights(seed=0)
    grid = (1, 32, 32)
    emb = e.vit_forward(e.patchify(torch.from_numpy(synthetic.tile_pixels(0)).cuda()), [grid])
    ids = synthetic.tile_prompt(geom, 0, grid)
    p, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], [grid], None)
    pos = p[:, 0].numpy()
    txt = [np.random.default_rng(40 + i).integers(0, 1000, 30 + i).astype(np.int64) for i in range(31)]
    tpos = [np.tile(np.arange(len(t)), (3, 1)) for t in txt]
    out = {}
    for B in (1, 5, 16, 17, 32):
        e.prefill([ids] + txt[: B - 1], [pos] + tpos[: B - 1], emb)
        g = e.decode(8, use_graph=True)
        e.prefill([ids] + txt[: B - 1], [pos] + tpos[: B - 1], emb)
        ea = e.decode(8, use_graph=False)
        assert torch.equal(g, ea), B
        out[B] = g[0].tolist()
    assert out[5] == out[16] and out[17] == out[32], out
    assert out[1][0] == out[5][0] == out[17][0]
    e.close()


def test_full_size_long_context_decode():
    """Full 3B geometry with a 2100-token prompt (decode attention beyond 2048 keys: scalar softmax path, second V^T sweep):
    graph == eager, and the decode logits of step k equal -- within the bf16 noise floor -- the all-position logits of the
    prompt extended by the k generated tokens (GEMV + decode attention vs GEMM + prefill attention)."""
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    geom = geometry_3b()
    e = Engine(geom, max_patches=256, max_prefill_tokens=2176, max_batch=1, max_ctx=2304, max_new_tokens=8)
    e.load_synthetic_weights(seed=0)
    ids = np.random.default_rng(9).integers(0, 150000, 2100).astype(np.int64)
    pos = np.tile(np.arange(len(ids)), (3, 1))
    e.prefill([ids], [pos], None)
    eager, trace = e.decode(8, trace=True, use_graph=False)
    e.prefill([ids], [pos], None)
    assert torch.equal(e.decode(8, use_graph=True), eager) and bool(torch.isfinite(trace).all())
    toks = eager[0].tolist()
    k = 4
    idk = np.concatenate([ids, np.asarray(toks[:k], dtype=np.int64)])
    allp = e.forward_logits([idk], [np.tile(np.arange(len(idk)), (3, 1))], None)
    for j in (0, k):                      # position 2099 + j predicts token j: compare with the decode-path logits of step j
        d = (allp[len(ids) - 1 + j] - trace[j, 0]).abs()
        scale = float(trace[j, 0].abs().max())
        assert float(d.max()) <= 0.08 * max(scale, 1.0), (j, float(d.max()), scale)
    e.close()
