"""PMC target: the MFMA GEMM alone on the LM gate/up and ViT qkv shapes at batch 32 -- run under rocprofv3 --pmc <counters>."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for M, N, K, epi in ((14336, 22016, 2048, 2), (32768, 3840, 1280, 0)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    ldo = N // 2 if epi == 2 else N
    out = torch.zeros(M, ldo, dtype=torch.bfloat16, device="cuda")
    for _ in range(4):
        assert L.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), ldo, None, None, None, epi, s) == 0
    torch.cuda.synchronize()
print("done")
