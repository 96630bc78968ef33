"""GPU micro-benchmark of the MFMA GEMM (sr_op_gemm) on the shapes the hot path launches."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [  # (name, M, N, K, epi)
    ("vit qkv  B32", 32768, 3840, 1280, 0), ("vit gateup B32", 32768, 6912, 1280, 2), ("vit down B32", 32768, 1280, 3456, 1),
    ("lm gateup B32", 14336, 22016, 2048, 2), ("lm down B32", 14336, 2048, 11008, 1), ("lm qkv B32", 14336, 2560, 2048, 0),
    ("vit qkv  B1", 1024, 3840, 1280, 0), ("vit gateup B1", 1024, 6912, 1280, 2), ("lm down B1", 448, 2048, 11008, 1),
    ("lm gateup B1", 448, 22016, 2048, 2), ("square 4096", 4096, 4096, 4096, 0), ("square 8192", 8192, 8192, 8192, 0),
]
for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    ldo = N // 2 if epi == 2 else N
    out = torch.zeros(M, ldo, dtype=torch.bfloat16, device="cuda")
    res = out if epi == 1 else None
    for _ in range(2):
        assert L.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), ldo, None, P(res), None, epi, s) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        L.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), ldo, None, P(res), None, epi, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    line = f"{name:16s} M={M:6d} N={N:6d} K={K:6d} epi={epi}: {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s"
    if name.startswith("lm"):      # the engine stores LM weights fragment-ordered: time that source layout too
        for _ in range(2):
            L.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), ldo, None, P(res), None, epi | 0x100, s)
        e0.record()
        for _ in range(reps):
            L.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), ldo, None, P(res), None, epi | 0x100, s)
        e1.record(); torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / reps
        line += f"   | tiled W: {ms2*1e3:9.1f} us  {2*M*N*K/ms2/1e9:8.1f} TF/s"
    print(line)
