"""A/B of the two launch forms of the 256-tile GEMM (gemm256.hip) on the batch-32 shapes of the hot path: persistent (one block per CU walks
its tiles, next tile's first units staged under the current tile's last k-tiles) against one tile per block (flag 0x2000)."""
import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [("vit qkv(store)", 32768, 3840, 1280, 0, 0), ("vit proj", 32768, 1280, 1280, 1, 0), ("vit gate/up", 32768, 6912, 1280, 2, 0),
          ("vit down", 32768, 1280, 3456, 1, 0), ("merger fc1", 8192, 5120, 5120, 3, 0), ("lm qkv(store)", 14336, 2560, 2048, 0, 0x100),
          ("lm o", 14336, 2048, 2048, 1, 0x100), ("lm gate/up", 14336, 22016, 2048, 2, 0x100), ("lm down", 14336, 2048, 11008, 1, 0x100)]
for name, M, N, K, epi, tl in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    ldo = N // 2 if epi == 2 else N
    out = torch.zeros(M, ldo, dtype=torch.bfloat16, device="cuda")
    res = out if epi == 1 else None
    row = {"shape": name, "M": M, "N": N, "K": K, "tiles": ((M + 255) // 256) * (N // 256)}
    for tag, fl in (("one_tile", 0x200 | 0x2000), ("persistent", 0x200), ("one_tile2", 0x200 | 0x2000), ("persistent2", 0x200)):
        for _ in range(3):
            assert L.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), ldo, None, P(res), None, epi | fl | tl, s) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            L.sr_op_gemm(P(a), K, P(w), M, N, K, P(out), ldo, None, P(res), None, epi | fl | tl, s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row[tag + "_us"] = round(ms * 1e3, 1)
        row[tag + "_TF"] = round(2 * M * N * K / ms / 1e9, 1)
    print(json.dumps(row), flush=True)
