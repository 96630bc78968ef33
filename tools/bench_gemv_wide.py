"""GPU micro-benchmark (round 3): decode GEMV launches at batch 32 with fragment-ordered x, default 4-wave blocks against the wide
8- / 16-wave blocks (SR_GEMV_W).  Weights distinct per repetition (no cache reuse), interleaved variants, one process."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(os.environ.get("PB", 32))
H, QN, I = 2048, 2560, 11008
R = 12
XT, OT, TL = 0x800, 0x1000, 0x100
x = torch.randn(32, I, device="cuda").to(torch.bfloat16)
bias = torch.zeros(QN, device="cuda").to(torch.bfloat16)
def mk(n, k): return (torch.randn(R, n, k, device="cuda") * 0.02).to(torch.bfloat16)
cases = [
    ("qkv bias", mk(QN, H), lambda w, o, ks: L.sr_op_gemv_fused(P(x), I, P(w), B, QN, H, P(o), QN, 3 | TL | XT, P(bias), None, C.c_float(0), None, 0, None, None, None, s), torch.zeros(32, QN, dtype=torch.bfloat16, device="cuda"), QN * H * 2, [1]),
    ("o resid", mk(H, H), lambda w, o, ks: L.sr_op_gemv_fused(P(x), I, P(w), B, H, H, P(o), H, 4 | TL | XT, None, None, C.c_float(0), None, 0, None, None, None, s), torch.zeros(32, H, dtype=torch.bfloat16, device="cuda"), H * H * 2, [1]),
    ("gate/up swiglu", mk(2 * I, H), lambda w, o, ks: L.sr_op_gemv_fused(P(x), I, P(w), B, 2 * I, H, P(o), I, 1 | TL | XT | OT, None, None, C.c_float(0), None, 0, None, None, None, s), torch.zeros(32, I, dtype=torch.bfloat16, device="cuda"), 2 * I * H * 2, [1]),
    ("down partial", mk(H, I), lambda w, o, ks: L.sr_op_gemv(P(x), I, P(w), B, H, I, P(o), ks, 0 | TL | XT, s), torch.zeros(4, 32, H, device="cuda"), H * I * 2, [4, 1]),
    ("down resid (K=11008)", mk(H, I), lambda w, o, ks: L.sr_op_gemv_fused(P(x), I, P(w), B, H, I, P(o), H, 4 | TL | XT, None, None, C.c_float(0), None, 0, None, None, None, s), torch.zeros(32, H, dtype=torch.bfloat16, device="cuda"), H * I * 2, [1]),
]
for name, W, fn, out, nbytes, kss in cases:
    for ks in kss:
        for wv in ("0", "8", "16"):
            os.environ["SR_GEMV_W"] = wv
            ok = all(fn(W[r], out, ks) == 0 for r in range(R))
            if not ok:
                print(f"B={B} {name:22s} ks={ks} W={wv:2s}: n/a")
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(5):
                for r in range(R): fn(W[r], out, ks)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (5 * R)
            print(f"B={B} {name:22s} ks={ks} W={wv:2s}: {us:7.2f} us  {nbytes/us/1e6:6.2f} TB/s")
