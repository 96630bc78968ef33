"""GPU probe: k_sample time per launch at the 3B vocabulary, full scan vs the LM-head block-maxima shortcut."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
V = 151936
for B in (1, 4, 32):
    lg = torch.randn(B, V, device="cuda") * 2
    bm = lg.view(B, V // 64, 64).amax(-1).contiguous()
    out = torch.zeros(B, dtype=torch.int64, device="cuda")
    for name, args in (("full scan", (None, 0, 0)), ("block maxima", (P(bm), V // 64, 64))):
        call = lambda: L.sr_op_sample(P(lg), B, V, C.c_float(1.0), 100, C.c_float(0.8), C.c_float(1.0), None, 1, None, P(out), args[0], args[1], args[2], s)
        for _ in range(3): assert call() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): call()
        b.record(); torch.cuda.synchronize()
        print(f"k_sample B={B} {name:13s}: {a.elapsed_time(b)*1000/50:.1f} us")
