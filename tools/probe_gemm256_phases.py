"""Which of the four phases of a k-tile does the 256 x 256 GEMM's k loop wait in?  Timing build (-DSR_G256_TIMING): the first wave of each wave row
stamps s_memtime at the phase boundaries of k-tile 5.  Prints the median shader clocks per phase for both wave rows, per shape (a phase's 16 MFMAs
are 256 clocks of a wave; the two waves of a SIMD alternate, so 512 clocks per phase = the MFMA pipe saturated)."""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "socioreasoner_amd", "libsocior_timing.so"))
L.sr_op_gemm.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
L.sr_dbg_g256_phases.argtypes = [C.c_void_p]
L.sr_dbg_g256_times.argtypes = [C.c_void_p, C.c_int]
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, M, N, K, epi, tl in [("vit gate/up", 32768, 6912, 1280, 2, 0), ("vit qkv(store)", 32768, 3840, 1280, 0, 0), ("lm gate/up", 14336, 22016, 2048, 2, 0x100),
                               ("lm down", 14336, 2048, 11008, 1, 0x100), ("one round", 8192, 2048, 2048, 0, 0)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    ldo = N // 2 if epi == 2 else N
    out = torch.zeros(M, ldo, dtype=torch.bfloat16, device="cuda")
    for _ in range(8):
        assert L.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), ldo, None, P(out) if epi == 1 else None, None, epi | 0x200 | tl, s) == 0
    torch.cuda.synchronize()
    raw = np.zeros(1024 * 2 * 6 + 2048, np.int64)
    assert L.sr_dbg_g256_phases(raw.ctypes.data) == 0
    t = raw[:1024 * 2 * 6].reshape(1024, 2, 6)
    ck = raw[1024 * 2 * 6:].reshape(1024, 2)
    wall = np.zeros((1024, 6), np.int64)
    assert L.sr_dbg_g256_times(wall.ctypes.data, 1024) == 0
    nb = min(1024, ((M + 255) // 256) * (N // 256))
    d = np.diff(t[:nb, :, :5], axis=2).astype(np.float64)          # [block][wave row][phase]
    ok = (d > 0).all(axis=2) & (d < 20000).all(axis=2)
    nbk = min(1024, ((M + 255) // 256) * (N // 256))
    ghz = (ck[:nbk, 1] - ck[:nbk, 0]) / ((wall[:nbk, 2] - wall[:nbk, 1]) * 10.0)          # shader clocks per ns of the 100 MHz wall clock, over the k loop
    row = {"shape": name, "shader_clock_GHz_during_k_loop_p10_p50_p90": [round(float(np.percentile(ghz, q)), 3) for q in (10, 50, 90)], "k_tile_clocks_median": [round(float(np.median(d[:, r].sum(axis=1)[ok[:, r]])), 0) for r in range(2)]}
    for r in range(2):
        row[f"wave_row_{r}_phase_clocks_p0_p1_p2_p3"] = [round(float(np.median(d[:, r, ph][ok[:, r]])), 0) for ph in range(4)]
    print(json.dumps(row), flush=True)
