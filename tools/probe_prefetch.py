"""GPU probe: does warming L2 / Infinity Cache with a weight region ahead of the GEMV that streams it shorten that GEMV?
Per case: cold GEMV (distinct weights per rep), prefetch alone, prefetch + GEMV (linear and XCD-matched prefetch)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, H, QN, I = 1, 2048, 2560, 11008
R = 12
x = torch.randn(B, I, device="cuda").to(torch.bfloat16)
nw = torch.ones(H, device="cuda").to(torch.bfloat16)
def mk(n, k): return (torch.randn(R, n, k, device="cuda") * 0.02).to(torch.bfloat16)
def timeit(fn, reps=5):
    for r in range(R): fn(r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(reps):
        for r in range(R): fn(r)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * R)
cases = [
  ("o    resid", mk(H, H), lambda w, o: L.sr_op_gemv_fused(P(x), I, P(w), B, H, H, P(o), H, 4 | 0x100, None, None, C.c_float(0), None, 0, None, None, None, s), torch.zeros(B, H, dtype=torch.bfloat16, device="cuda"), 16 * H * 2),
  ("gate/up swiglu+norm", mk(2 * I, H), lambda w, o: L.sr_op_gemv_fused(P(x), I, P(w), B, 2 * I, H, P(o), I, 1 | 0x100, None, P(nw), C.c_float(1e-6), None, 0, None, None, None, s), torch.zeros(B, I, dtype=torch.bfloat16, device="cuda"), 32 * H * 2),
  ("down partial ks=2", mk(H, I), lambda w, o: L.sr_op_gemv(P(x), I, P(w), B, H, I, P(o), 2, 0 | 0x100, s), torch.zeros(4, B, H, device="cuda"), 0),
]
filler = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
for name, W, fn, out, tile_bytes in cases:
    full = W[0].numel() * 2
    cold = timeit(lambda r: fn(W[r], out))
    print(f"{name:22s} cold {cold:7.2f} us ({full/cold/1e6:5.2f} TB/s)")
    for frac in (1.0, 0.5, 0.25):
        nbytes = int(full * frac) // (1 << 16) * (1 << 16)
        for tb in ([0, tile_bytes] if tile_bytes else [0]):
            for blocks in (256, 1024):
                pf = timeit(lambda r: L.sr_op_prefetch(P(W[r]), nbytes, tb, blocks, s))
                both = timeit(lambda r: (L.sr_op_prefetch(P(W[r]), nbytes, tb, blocks, s), fn(W[r], out)))
                print(f"   prefetch {nbytes/1e6:6.1f} MB tile_bytes={tb:6d} blocks={blocks:4d}: alone {pf:6.2f} us; +gemv {both:7.2f} us -> gemv {both-pf:6.2f} us")
