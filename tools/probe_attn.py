"""GPU probe: phase timestamps of the fused decode-attention kernel at the bench shape (B=1, ctx 576)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib
L = lib.load()
B, HQ, HK, CTX = int(os.environ.get("PB", 1)), 16, 2, 640
P = lambda t: C.c_void_p(t.data_ptr())
qkv = torch.randn(B, (HQ + 2 * HK) * 128, device="cuda").to(torch.bfloat16)
pos = torch.full((B,), 600, dtype=torch.int32, device="cuda")
ctx = torch.full((B,), 576, dtype=torch.int32, device="cuda")
inv = (1.0 / (1e6 ** (torch.arange(0, 128, 2).float() / 128)))
ang = torch.arange(CTX + 1).float()[:, None] * inv[None]
rc, rs = ang.cos().to(torch.bfloat16).cuda(), ang.sin().to(torch.bfloat16).cuda()
kc = torch.randn(B, HK, CTX, 128, device="cuda").to(torch.bfloat16)
vc = torch.randn(B, HK, 128, CTX, device="cuda").to(torch.bfloat16)
out = torch.zeros(B, HQ * 128, dtype=torch.bfloat16, device="cuda")
dbg = torch.zeros(B * HQ * CTX, dtype=torch.bfloat16, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for it in range(5):
    assert L.sr_op_attn_decode(P(qkv), qkv.shape[1], P(pos), P(ctx), P(rc), P(rs), P(kc), P(vc), P(out), HQ * 128, B, HQ, HK, CTX,
                               C.c_float(128 ** -0.5), P(dbg), s) == 0
    torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200):
    L.sr_op_attn_decode(P(qkv), qkv.shape[1], P(pos), P(ctx), P(rc), P(rs), P(kc), P(vc), P(out), HQ * 128, B, HQ, HK, CTX, C.c_float(128 ** -0.5), P(dbg), s)
b.record(); torch.cuda.synchronize()
print("avg per launch (back-to-back, us):", a.elapsed_time(b) * 1000 / 200)
