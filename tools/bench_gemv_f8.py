"""GPU micro-benchmark of the fp8-weight decode GEMV launches of one LM layer (weights distinct per rep: no cache reuse)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(os.environ.get("PB", 1))
H, QN, I = 2048, 2560, 11008
R = 12
x = torch.randn(B, I, device="cuda").to(torch.bfloat16)
nw = torch.ones(H, device="cuda").to(torch.bfloat16)
bias = torch.zeros(QN, device="cuda").to(torch.bfloat16)
sc = torch.ones(2 * I, device="cuda")
fused = B <= 4
eps = C.c_float(1e-6)
def mk(n, k): return torch.empty(R, n * k, dtype=torch.uint8, device="cuda").random_(0, 120)
cases = [
  ("qkv bias" + ("+norm" if fused else ""), mk(QN, H), lambda w, o: L.sr_op_gemv_f8(P(x), I, P(w), P(sc), B, QN, H, P(o), QN, 3, P(bias), P(nw) if fused else None, eps, 1, s), torch.zeros(B, QN, dtype=torch.bfloat16, device="cuda"), QN * H),
  ("o resid", mk(H, H), lambda w, o: L.sr_op_gemv_f8(P(x), I, P(w), P(sc), B, H, H, P(o), H, 4, None, None, eps, 1, s), torch.zeros(B, H, dtype=torch.bfloat16, device="cuda"), H * H),
  ("gate/up swiglu" + ("+norm" if fused else ""), mk(2 * I, H), lambda w, o: L.sr_op_gemv_f8(P(x), I, P(w), P(sc), B, 2 * I, H, P(o), I, 1, None, P(nw) if fused else None, eps, 1, s), torch.zeros(B, I, dtype=torch.bfloat16, device="cuda"), 2 * I * H),
  ("down partial ks=2", mk(H, I), lambda w, o: L.sr_op_gemv_f8(P(x), I, P(w), P(sc), B, H, I, P(o), H, 0, None, None, eps, 2, s), torch.zeros(4, B, H, device="cuda"), H * I),
  ("down partial ks=4", mk(H, I), lambda w, o: L.sr_op_gemv_f8(P(x), I, P(w), P(sc), B, H, I, P(o), H, 0, None, None, eps, 4, s), torch.zeros(4, B, H, device="cuda"), H * I),
]
for name, W, fn, out, nbytes in cases:
    for r in range(R): assert fn(W[r], out) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(5):
        for r in range(R): fn(W[r], out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * R)
    print(f"B={B} f8 {name:24s}: {us:7.2f} us  {nbytes/us/1e6:6.2f} TB/s")
