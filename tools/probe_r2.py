"""PMC targets of round 2 (run under rocprofv3 --pmc ...): `gemv` = the batch-32 decode GEMV family on fragment-ordered activations
(gate/up, 4-slab down-projection, LM head) and the batch-1 family; `gemm` = the 256 x 256 8-phase GEMM on the LM gate/up and ViT qkv
shapes at batch 32."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
what = sys.argv[1]
if what == "gemm":
    for M, N, K, epi, tiled in ((14336, 22016, 2048, 2, 0x100), (32768, 3840, 1280, 0, 0), (14336, 2048, 11008, 1, 0x100)):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
        ldo = N // 2 if epi == 2 else N
        out = torch.zeros(M, ldo, dtype=torch.bfloat16, device="cuda")
        for _ in range(4):
            assert L.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), ldo, None, P(out) if epi == 1 else None, None, epi | tiled | 0x200, s) == 0
        torch.cuda.synchronize()
elif len(sys.argv) > 2 and sys.argv[2] == "fp8":
    # round 5: the fp8 weight stream of the same launches (BASELINE.json configs[4]; the LM head stays bf16): gate/up and the down-projection at 32 rows and at 1
    H, I = 2048, 11008
    g8 = torch.randint(0, 120, (2 * I * H,), dtype=torch.uint8, device="cuda")
    d8 = torch.randint(0, 120, (H * I,), dtype=torch.uint8, device="cuda")
    sc = torch.ones(2 * I, device="cuda")
    for B in (32, 1):
        XT, OT, ks = (0x800, 0x1000, 4) if B > 4 else (0, 0, 2)
        x = torch.randn(32, I, device="cuda").to(torch.bfloat16)
        act = torch.zeros(32, I, dtype=torch.bfloat16, device="cuda")
        part = torch.zeros(4, B, H, device="cuda")
        nw = torch.ones(H, dtype=torch.bfloat16, device="cuda")
        eps = C.c_float(1e-6)
        for _ in range(5):
            assert L.sr_op_gemv_f8(P(x), H, P(g8), P(sc), B, 2 * I, H, P(act), I, 1 | XT | OT, None, P(nw) if B <= 4 else None, eps, 1, s) == 0     # 45.1 MB
            assert L.sr_op_gemv_f8(P(act), I, P(d8), P(sc), B, H, I, P(part), H, 0 | XT, None, None, eps, ks, s) == 0                                # 22.5 MB
        torch.cuda.synchronize()
else:
    H, I, V = 2048, 11008, 151936
    wg = (torch.randn(2 * I, H, device="cuda") * 0.02).to(torch.bfloat16)
    wd = (torch.randn(H, I, device="cuda") * 0.02).to(torch.bfloat16)
    wv = (torch.randn(V, H, device="cuda") * 0.02).to(torch.bfloat16)
    for B in (32, 1):
        XT, OT, ks = (0x800, 0x1000, 4) if B > 4 else (0, 0, 2)
        x = torch.randn(32, I, device="cuda").to(torch.bfloat16)
        act = torch.zeros(32, I, dtype=torch.bfloat16, device="cuda")
        part = torch.zeros(4, B, H, device="cuda")
        lg = torch.zeros(B, V, device="cuda")
        nb = L.sr_op_gemv_f32_blocks(V, B, H, 0)
        av = torch.zeros(B, nb, device="cuda")
        ai = torch.zeros(B, nb, dtype=torch.int32, device="cuda")
        eps = C.c_float(1e-6)
        for _ in range(5):
            L.sr_op_gemv_fused(P(x), H, P(wg), B, 2 * I, H, P(act), I, 1 | 0x100 | XT | OT, None, None, eps, None, 0, None, None, None, s)   # 90.2 MB
            L.sr_op_gemv(P(act), I, P(wd), B, H, I, P(part), ks, 0 | 0x100 | XT, s)                                                        # 45.1 MB
            L.sr_op_gemv_fused(P(x), H, P(wv), B, V, H, P(lg), V, 2 | 0x100 | XT, None, None, eps, None, 0, None, P(av), P(ai), s)         # 622.3 MB
        torch.cuda.synchronize()
print("done")
