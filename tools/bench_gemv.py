"""GPU micro-benchmark of the decode GEMV launches of one LM layer + LM head (weights distinct per rep: no cache reuse)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(os.environ.get("PB", 1))
H, QN, I, V = 2048, 2560, 11008, 151936
R = int(os.environ.get("PR", 12))
x = torch.randn(B, I, device="cuda").to(torch.bfloat16)
nw = torch.ones(H, device="cuda").to(torch.bfloat16)
bias = torch.zeros(QN, device="cuda").to(torch.bfloat16)
slabs = torch.zeros(2, B, H, device="cuda")
xo = torch.zeros(B, H, dtype=torch.bfloat16, device="cuda")
fused = B <= 4
def mk(n, k): return (torch.randn(R, n, k, device="cuda") * 0.02).to(torch.bfloat16)
cases = [
  ("qkv  bias" + ("+norm+slabs" if fused else ""), mk(QN, H), lambda w, o: L.sr_op_gemv_fused(P(x), I, P(w), B, QN, H, P(o), QN, 3 | 0x100, P(bias), P(nw) if fused else None, C.c_float(1e-6), P(slabs) if fused else None, 2 if fused else 0, P(xo) if fused else None, None, None, s), torch.zeros(B, QN, dtype=torch.bfloat16, device="cuda"), QN * H * 2),
  ("o    resid", mk(H, H), lambda w, o: L.sr_op_gemv_fused(P(x), I, P(w), B, H, H, P(o), H, 4 | 0x100, None, None, C.c_float(0), None, 0, None, None, None, s), torch.zeros(B, H, dtype=torch.bfloat16, device="cuda"), H * H * 2),
  ("gate/up swiglu" + ("+norm" if fused else ""), mk(2 * I, H), lambda w, o: L.sr_op_gemv_fused(P(x), I, P(w), B, 2 * I, H, P(o), I, 1 | 0x100, None, P(nw) if fused else None, C.c_float(1e-6), None, 0, None, None, None, s), torch.zeros(B, I, dtype=torch.bfloat16, device="cuda"), 2 * I * H * 2),
]
nb = L.sr_op_gemv_f32_blocks(V, B, H, 1 if fused else 0)
av = torch.zeros(B, 2400, device="cuda"); ai = torch.zeros(B, 2400, dtype=torch.int32, device="cuda")
WV = (torch.randn(2, V, H, device="cuda") * 0.02).to(torch.bfloat16)
cases.append(("lm_head f32+argmax", WV, lambda w, o: L.sr_op_gemv_fused(P(x), I, P(w), B, V, H, P(o), V, 2 | 0x100, None, P(nw) if fused else None, C.c_float(1e-6), None, 0, None, P(av), P(ai), s), torch.zeros(B, V, device="cuda"), V * H * 2))
for ks in (2, 4):
    cases.append((f"down partial ks={ks}", mk(H, I), (lambda ks: lambda w, o: L.sr_op_gemv(P(x), I, P(w), B, H, I, P(o), ks, 0 | 0x100, s))(ks), torch.zeros(4, B, H, device="cuda"), H * I * 2))
for name, W, fn, out, nbytes in cases:
    RR = W.shape[0]
    for r in range(RR): assert fn(W[r], out) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(5):
        for r in range(RR): fn(W[r], out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * RR)
    print(f"B={B} G32={os.environ.get('SR_GEMV32','1')} KP={os.environ.get('SR_GEMV_KP','dflt')} {name:28s}: {us:7.2f} us  {nbytes/us/1e6:6.2f} TB/s")
