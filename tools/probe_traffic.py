"""PMC target: the decode GEMV weight stream alone (no graphs, no engine) -- run under rocprofv3 --pmc FETCH_SIZE."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, H, I, V = 1, 2048, 11008, 151936
x = torch.randn(B, I, device="cuda").to(torch.bfloat16)
wg = (torch.randn(2 * I, H, device="cuda") * 0.02).to(torch.bfloat16)
wd = (torch.randn(H, I, device="cuda") * 0.02).to(torch.bfloat16)
wv = (torch.randn(V, H, device="cuda") * 0.02).to(torch.bfloat16)
act = torch.zeros(B, I, dtype=torch.bfloat16, device="cuda")
part = torch.zeros(2, B, H, device="cuda")
lg = torch.zeros(B, V, device="cuda")
for _ in range(5):
    L.sr_op_gemv(P(x), I, P(wg), B, 2 * I, H, P(act), 1, 1, s)      # gate/up: 90.2 MB of weights
    L.sr_op_gemv(P(act), I, P(wd), B, H, I, P(part), 2, 0, s)       # down: 45.1 MB
    L.sr_op_gemv(P(x), I, P(wv), B, V, H, P(lg), 1, 2, s)           # LM head: 622.3 MB
torch.cuda.synchronize()
print("done")
