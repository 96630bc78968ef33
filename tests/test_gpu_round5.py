"""Round 5 GPU tests (through the C ABI): the RMSNorms of a 5..32-row decode layer inside the GEMV launches that produce their input
(csrc/rownorm.h), the fp8 weight stream above 32 rows, the library's switches read once."""
import numpy as np
import pytest
import torch

from tests.util import switch

pytestmark = pytest.mark.gpu


def _prompts(rng, n, lo=5, hi=40, vocab=2000):
    ids = [rng.integers(0, vocab, int(rng.integers(lo, hi))).astype(np.int64) for _ in range(n)]
    pos = [np.tile(np.arange(len(x)), (3, 1)).astype(np.int64) for x in ids]
    return ids, pos


# ------------------------------------------------------------------------------------------------ in-launch RMSNorm (VERDICT round 4, R1 / next #1a)
@pytest.mark.parametrize("fp8", [False, True])
def test_tail_rmsnorm_equals_the_rmsnorm_launches_tiny(monkeypatch, fp8):
    """At 5..32 batch rows the o_proj (RESID) and down-projection (PARTIAL) GEMVs finish their rows as the next launch's normalised x: the
    last B blocks to arrive (arrival tickets, write-through stores, sc1 loads: the MI355X guide's split-K counter recipe) add the float32
    slabs to the residual stream and apply the RMSNorm, one row per block, with the loads, float32 association and reduction order of
    k_rmsnorm_row; the first layer's norm rides in k_step.  SR_TAIL_NORM=0 keeps the RMSNorm launches.  Same bits: every logit of every
    step at 5 / 16 / 17 / 32 rows (both GEMV kernels, both slab counts), eager and graph-replayed, and no tail block ever gave up waiting."""
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.engine import Engine
    geom = geometry_tiny()
    rng = np.random.default_rng(5)
    ids, pos = _prompts(rng, 32)
    out = {}
    for flag in ("0", "3"):
        switch(monkeypatch, "SR_TAIL_NORM", flag)
        e = Engine(geom, max_patches=64, max_prefill_tokens=64 * 32, max_batch=32, max_ctx=128, max_new_tokens=16, lm_fp8=fp8)
        e.load_synthetic_weights(seed=0)
        res = []
        for B in (5, 16, 17, 32):
            e.prefill(ids[:B], pos[:B])
            toks, tr = e.decode(12, trace=True, use_graph=False)
            e.prefill(ids[:B], pos[:B])
            toks_g, tr_g = e.decode(12, trace=True, use_graph=True)
            assert torch.equal(toks, toks_g) and torch.equal(tr, tr_g), (flag, B)
            for _ in range(3):          # replays: run-to-run identical bits (a tail that read a row before its last piece landed would differ)
                e.prefill(ids[:B], pos[:B])
                t2, tr2 = e.decode(12, trace=True, use_graph=True)
                assert torch.equal(tr2, tr), (flag, B)
            res.append((toks.clone(), tr.clone()))
        assert e.tail_timeouts() == 0
        out[flag] = res
        e.close()
    for (t0, r0), (t1, r1) in zip(out["0"], out["3"]):
        assert torch.equal(t0, t1)
        assert torch.equal(r0, r1), float((r0 - r1).abs().max())


def test_tail_rmsnorm_full_size_32_rows_and_continuous_batching(monkeypatch):
    """The same at the 3B geometry (8 layers deep: N = 2048 -> 128 o_proj blocks / 256 down-projection blocks for 32 tail rows, 4 slabs): 32
    ragged prompts decoded 24 steps with and without the tails -- bit-identical logits; then the rows of the tail engine served by the
    continuous batcher with the next admission staged on CU-masked streams UNDER the decode steps (uneven load on the chip while the tails
    poll and read) -- every request's tokens equal the static run's."""
    from dataclasses import replace
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    from socioreasoner_amd.serving import ContinuousBatcher, Request
    geom = geometry_3b()
    geom = replace(geom, text=replace(geom.text, num_hidden_layers=8), vision=replace(geom.vision, depth=2, fullatt_block_indexes=(1,)))
    rng = np.random.default_rng(11)
    ids, pos = _prompts(rng, 64, lo=20, hi=120, vocab=150000)
    out = {}
    for flag in ("0", "3"):
        switch(monkeypatch, "SR_TAIL_NORM", flag)
        e = Engine(geom, max_patches=64, max_prefill_tokens=128 * 32, max_batch=32, max_ctx=256, max_new_tokens=24, kv_slots=64)
        e.load_synthetic_weights(seed=0)
        e.prefill(ids[:32], pos[:32])
        toks, tr = e.decode(24, trace=True, use_graph=True)
        out[flag] = (toks.clone(), tr.clone())
        if flag == "3":
            for _ in range(4):
                e.prefill(ids[:32], pos[:32])
                _, tr2 = e.decode(24, trace=True, use_graph=True)
                assert torch.equal(tr2, tr)
            cb = ContinuousBatcher(e, eos=[], pad_id=0, steps_per_poll=4, overlap=True)
            got = cb.run([Request(ids=ids[i], pos3=pos[i], max_new=24, tag=i) for i in range(64)])
            for i in range(32):
                assert got[i] == toks[i].tolist(), i
            assert e.tail_timeouts() == 0
        e.close()
    assert torch.equal(out["0"][0], out["3"][0])
    assert torch.equal(out["0"][1], out["3"][1]), float((out["0"][1] - out["3"][1]).abs().max())


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
    # the weight pre-split on the host into three bf16 planes (| 0x2000: what socioreasoner_amd/sam2.py hands the kernel): the same bits as the in-kernel split
    hi = dW.to(torch.bfloat16)
    r1 = dW - hi.float()
    mid = r1.to(torch.bfloat16)
    w3 = torch.stack([hi, mid, (r1 - mid.float()).to(torch.bfloat16)]).contiguous()
    out3 = torch.zeros(M, N, device="cuda")
    assert L.sr_op_gemm_f32(_P(dA), K, _P(w3), M, N, K, _P(out3), N, _P(db), None, None, 0x2000, _sp()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out3.cpu(), outs["1"])
    ulp = float(want.abs().max()) * 2.0 ** -23
    rel = float(((outs["1"] - outs["0"]).abs() / (want.abs().float() + float(want.abs().mean()))).max())
    _record("r05_split_gemm.json", f"{M}x{N}x{K}", {"max_abs_err_f32_mfma": errs["0"], "max_abs_err_split_bf16": errs["1"], "ulp_of_largest_output": ulp,
                                                   "max_rel_diff_between_kernels": rel})
    assert errs["1"] <= 1.25 * errs["0"] + ulp, errs
    assert rel < 1e-4, rel
