#!/usr/bin/env python3
"""SAM2's float32 GEMM at Hiera-L's shapes (8 tiles per encoder pass): the f32-input MFMA kernel (SR_SAM_F32_SPLIT=0) against the three-term
bf16 split on the bf16 matrix pipe (default).  One JSON line per shape and kernel: us per launch, TFLOP/s (2 M N K), fraction of the 157.3 TF/s
float32 MFMA peak."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib  # noqa: E402

L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
# (tokens of 8 tiles, N, K) of the Hiera-L stages: qkv / proj / mlp of stages 1-4 (embed dims 144, 288, 576, 1152; 65536 .. 1024 tokens per tile)
SHAPES = [(524288, 432, 144), (524288, 576, 144), (131072, 864, 288), (131072, 1152, 288), (32768, 1728, 576), (32768, 2304, 576), (32768, 576, 2304),
          (8192, 3456, 1152), (8192, 4608, 1152), (8192, 1152, 4608)]
for M, N, K in SHAPES:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    o = torch.empty(M, N, device="cuda")
    for flag, name in (("0", "f32-input MFMA"), ("1", "split-bf16 x3, both operands split in the kernel")):
        os.environ["SR_SAM_F32_SPLIT"] = flag
        lib.reload_switches()
        call = lambda: L.sr_op_gemm_f32(P(a), K, P(w), M, N, K, P(o), N, None, None, None, 0, s)
        for _ in range(2):
            assert call() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        tf = 2.0 * M * N * K / us / 1e6
        print(json.dumps({"M": M, "N": N, "K": K, "kernel": name, "us": round(us, 1), "TFLOPs": round(tf, 1), "frac_of_f32_mfma_peak": round(tf / 157.3, 3)}), flush=True)
    del a, w, o
