#!/usr/bin/env python3
"""Op-level timing on the GPU box (HIP events on the launch stream, random operands, interleaved variants -- MI355X guide
rules 24/25).  Usage:
    python tools/bench_ops.py gemm            # hot-path GEMM shapes at batch 32: 128-tile kernel vs 256-tile 8-phase kernel
    python tools/bench_ops.py gemv [M ...]    # the decode step's weight-streaming launches at batch M (default 1 and 32)
Prints one JSON object per line; nothing here is part of the product path."""
from __future__ import annotations

import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from socioreasoner_amd import lib as L  # noqa: E402

EPI = {"store": 0, "resid": 1, "swiglu": 2, "gelu": 3, "f32": 4}
TILED, F256, F128 = 0x100, 0x200, 0x400


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def time_variants(fns, reps=12, rounds=3):
    """interleaved rounds; returns the median ms per call of every variant"""
    res = [[] for _ in fns]
    for f in fns:
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in enumerate(fns):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                f()
            b.record()
            torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / reps)
    return [sorted(r)[len(r) // 2] for r in res]


def bench_gemm():
    lib = L.load()
    shapes = [("vit qkv", 32768, 3840, 1280, "store", False), ("vit proj", 32768, 1280, 1280, "resid", False),
              ("vit gate/up", 32768, 6912, 1280, "swiglu", False), ("vit down", 32768, 1280, 3456, "resid", False),
              ("merger fc1", 8192, 5120, 5120, "gelu", False), ("merger fc2", 8192, 2048, 5120, "store", False),
              ("lm qkv", 14336, 2560, 2048, "store", True), ("lm o", 14336, 2048, 2048, "resid", True),
              ("lm gate/up", 14336, 22016, 2048, "swiglu", True), ("lm down", 14336, 2048, 11008, "resid", True),
              ("lm qkv b8", 3584, 2560, 2048, "store", True), ("lm gate/up b8", 3584, 22016, 2048, "swiglu", True),
              ("vit qkv b8", 8192, 3840, 1280, "store", False), ("vit proj b8", 8192, 1280, 1280, "resid", False)]
    for name, M, N, K, epi, tiled in shapes:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        b = (torch.randn(N, device="cuda") * 0.1).to(torch.bfloat16)
        No = N // 2 if epi == "swiglu" else N
        out = torch.zeros(M, No, dtype=torch.bfloat16, device="cuda")
        res = out if epi == "resid" else None

        def mk(force):
            flags = EPI[epi] | force | (TILED if tiled else 0)
            return lambda: lib.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), No, P(b), P(res), None, flags, stream())
        t128, t256 = time_variants([mk(F128), mk(F256)])
        fl = 2.0 * M * N * K
        print(json.dumps({"op": "gemm", "shape": name, "M": M, "N": N, "K": K, "epi": epi, "tiled": tiled,
                          "ms_128": round(t128, 4), "ms_256": round(t256, 4), "TF_128": round(fl / t128 / 1e9, 1), "TF_256": round(fl / t256 / 1e9, 1)}), flush=True)


def bench_gemm_mx():
    """LM prefill shapes at batch 32: bf16 256-tile GEMM vs the fp8 x fp8 block-scaled (MX) GEMM, plus the activation quantiser pass"""
    lib = L.load()
    for name, M, N, K, epi in [("lm qkv", 14336, 2560, 2048, "store"), ("lm o", 14336, 2048, 2048, "resid"), ("lm gate/up", 14336, 22016, 2048, "swiglu"),
                               ("lm down", 14336, 2048, 11008, "resid"), ("lm gate/up 896", 38912, 22016, 2048, "swiglu"), ("lm down 896", 38912, 2048, 11008, "resid")]:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
        w8 = torch.randint(0, 120, (N, K), dtype=torch.uint8, device="cuda")
        wsc = torch.ones(N, dtype=torch.float32, device="cuda")
        b = (torch.randn(N, device="cuda") * 0.1).to(torch.bfloat16)
        No = N // 2 if epi == "swiglu" else N
        out = torch.zeros(M, No, dtype=torch.bfloat16, device="cuda")
        res = out if epi == "resid" else None
        rp = (M + 255) // 256 * 256
        q = torch.zeros(M, K, dtype=torch.uint8, device="cuda")
        sc = torch.zeros(K // 128, rp, 4, dtype=torch.uint8, device="cuda")
        f_bf = lambda: lib.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), No, P(b), P(res), None, EPI[epi] | F256 | TILED, stream())
        f_q = lambda: lib.sr_op_quant_mx(P(a), K, M, K, P(q), P(sc), rp, stream())
        f_mx = lambda: lib.sr_op_gemm_mx(P(q), K, P(sc), rp, P(w8), P(wsc), M, N, K, P(out), No, P(b), P(res), EPI[epi], stream())
        f_q()
        t_bf, t_mx, t_q = time_variants([f_bf, f_mx, f_q])
        fl = 2.0 * M * N * K
        print(json.dumps({"op": "gemm_mx", "shape": name, "M": M, "N": N, "K": K, "ms_bf16": round(t_bf, 4), "ms_mx": round(t_mx, 4), "ms_quant": round(t_q, 4),
                          "TF_bf16": round(fl / t_bf / 1e9, 1), "TF_mx": round(fl / t_mx / 1e9, 1), "TF_mx_incl_quant": round(fl / (t_mx + t_q) / 1e9, 1)}), flush=True)


def bench_gemv(Ms):
    lib = L.load()
    H, QN, I, V = 2048, 2560, 11008, 151936
    nl = 12                                      # distinct weight copies: nothing is served from L2 / Infinity Cache between launches
    dev = "cuda"
    wq = torch.empty(nl, QN, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
    wo = torch.empty(nl, H, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
    wg = torch.empty(nl, 2 * I, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
    wd = torch.empty(nl, H, I, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
    wv = torch.empty(2, V, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
    eps = C.c_float(1e-6)
    TL = 0x100
    for M in Ms:
        part = torch.empty(8, M, QN, dtype=torch.float32, device=dev)
        nw = torch.ones(H, dtype=torch.bfloat16, device=dev)
        bq = torch.zeros(QN, dtype=torch.bfloat16, device=dev)
        qkv_o = torch.empty(M, QN, dtype=torch.bfloat16, device=dev)
        xr = torch.zeros(M, H, dtype=torch.bfloat16, device=dev)
        fused = M <= 4
        ksd = int(os.environ.get("SR_BENCH_KSD", "4" if M > 16 else "2"))
        s = stream()
        # batches > 4: activations travel in fragment order (engine.hip x_tiled); the values are random either way.  x / act are
        # over-allocated to 32 rows so that the tiled addressing of a 16-row group stays in bounds
        XT = 0x800 if (M > 4 and os.environ.get("SR_XTILED", "1") != "0") else 0
        OT = 0x1000 if XT else 0
        x = torch.empty(32, I, dtype=torch.bfloat16, device=dev).normal_(0, 1)
        act = torch.empty(32, I, dtype=torch.bfloat16, device=dev).normal_(0, 1)
        nbv = lib.sr_op_gemv_f32_blocks(V, M, H, 1 if fused else 0)
        av = torch.empty(M, nbv, dtype=torch.float32, device=dev)
        ai = torch.empty(M, nbv, dtype=torch.int32, device=dev)
        lg = torch.empty(M, V, dtype=torch.float32, device=dev)
        ops = {
            "qkv": (lambda l: lib.sr_op_gemv_fused(P(x), H, P(wq[l]), M, QN, H, P(qkv_o), QN, 3 | TL | XT, P(bq), P(nw) if fused else None, eps, None, 0, None, None, None, s), QN * H * 2),
            "o": (lambda l: lib.sr_op_gemv_fused(P(x), H, P(wo[l]), M, H, H, P(xr), H, 4 | TL | XT, None, None, eps, None, 0, None, None, None, s), H * H * 2),
            "gate_up": (lambda l: lib.sr_op_gemv_fused(P(x), H, P(wg[l]), M, 2 * I, H, P(act), I, 1 | TL | XT | OT, None, P(nw) if fused else None, eps, None, 0, None, None, None, s), 2 * I * H * 2),
            "down": (lambda l: lib.sr_op_gemv(P(act), I, P(wd[l]), M, H, I, P(part), ksd, 0 | TL | XT, s), H * I * 2),
            "head": (lambda l: lib.sr_op_gemv_fused(P(x), H, P(wv[l % 2]), M, V, H, P(lg), V, 2 | TL | XT, None, P(nw) if fused else None, eps, None, 0, None, P(av), P(ai), s), V * H * 2),
        }
        for name, (fn, nbytes) in ops.items():
            # one hipGraph of nl launches of this op (distinct weights each): replayed, so that the host launch cost (ctypes +
            # hipLaunchKernel, ~5 us) does not bound kernels that take about as long
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                s = stream()
                ops_l = dict(ops)          # lambdas read `s` at call time
                for l in range(nl):
                    fn(l)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    s = stream()
                    for l in range(nl):
                        fn(l)
                g.replay()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(5):
                    g.replay()
                b.record()
                torch.cuda.synchronize()
            us = a.elapsed_time(b) / (5 * nl) * 1e3
            print(json.dumps({"op": "gemv", "name": name, "M": M, "us": round(us, 2), "GBs": round(nbytes / us / 1e3, 1),
                              "variant": os.environ.get("SR_GEMV_VARIANT", "default")}), flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
    if what == "gemm":
        bench_gemm()
    elif what == "gemm_mx":
        bench_gemm_mx()
    else:
        bench_gemv([int(a) for a in sys.argv[2:]] or [1, 32])
