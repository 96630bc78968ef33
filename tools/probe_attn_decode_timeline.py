"""Where do the two decode attention launches (k_attn_dec_scores, k_attn_dec_pv) spend their 5 + 8 us at 32 rows?  Timing build of attention.hip
(-DSR_ATTN_TIMING, socioreasoner_amd/libsocior_timing.so; recipe in tools/experiments/README.md): thread 0 of every block stamps the 100 MHz
clock at the kernels' phase boundaries.  32 sequences x 2 kv heads, 450 cached tokens, the 3B LM's head layout; KV caches rotate through R
copies so that they come from HBM like in the engine (36 layers x 15 MB do not stay in L2)."""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "socioreasoner_amd", "libsocior_timing.so"))
vp, ci = C.c_void_p, C.c_int
L.sr_op_attn_decode.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, C.c_float, vp, vp]
L.sr_dbg_attn_dec_times.argtypes = [vp]
P = lambda t: vp(t.data_ptr())
s = vp(torch.cuda.current_stream().cuda_stream)
B, HQ, HK, HD, CTX, N, R = 32, 16, 2, 128, 512, 450, 24
qkv = (torch.randn(B, (HQ + 2 * HK) * HD, device="cuda") * 1.5).to(torch.bfloat16)
kc = [torch.randn(B, HK, CTX, HD, device="cuda").to(torch.bfloat16) for _ in range(R)]
vc = [torch.randn(B, HK, HD, CTX, device="cuda").to(torch.bfloat16) for _ in range(R)]
ctx = torch.full((B,), N, dtype=torch.int32, device="cuda")
pos = torch.full((B,), N + 3, dtype=torch.int32, device="cuda")
inv = 1.0 / (1e6 ** (torch.arange(0, HD, 2).float() / HD))
ang = torch.arange(CTX + 8).float()[:, None] * inv[None]
rc, rs = ang.cos().to(torch.bfloat16).cuda(), ang.sin().to(torch.bfloat16).cuda()
out = torch.zeros(B, HQ * HD, dtype=torch.bfloat16, device="cuda")
scratch = torch.zeros(B * HQ * CTX, dtype=torch.bfloat16, device="cuda")
run = lambda r: L.sr_op_attn_decode(P(qkv), qkv.shape[1], P(pos), P(ctx), P(rc), P(rs), P(kc[r]), P(vc[r]), P(out), HQ * HD, B, HQ, HK, CTX, C.c_float(HD ** -0.5), P(scratch), s)
for it in range(3):
    for r in range(R):
        assert run(r) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(5):
    for r in range(R):
        run(r)
e1.record(); torch.cuda.synchronize()
pair_us = e0.elapsed_time(e1) * 1e3 / (5 * R)
t = np.zeros((2, 8192, 5), np.int64)
assert L.sr_dbg_attn_dec_times(t.ctypes.data) == 0
names = [("k_attn_dec_scores", ["state + q/k/v + rotary row arrived", "q / k rotated (LDS)", "K fragments arrived + MFMA", "scores stored"]),
         ("k_attn_dec_pv", ["state + score rows arrived, in LDS", "softmax", "V^T arrived + MFMA", "reduce + store"])]
print(json.dumps({"pair_us_back_to_back": round(pair_us, 2), "rows": B, "kv_heads": HK, "cached_tokens": N}))
for k, (name, phases) in enumerate(names):
    tt = t[k]
    newest = tt[:, 4].max()
    live = (tt[:, 0] > newest - 3000) & (tt[:, 4] >= tt[:, 0]) & (tt[:, 1] > 0)
    x = tt[live].astype(np.float64) * 0.01
    t0 = x[:, 0].min()
    med = lambda v: round(float(np.median(v)), 2)
    row = {"kernel": name, "blocks_that_ran_to_the_end": int(live.sum()), "span_us": round(float(x[:, 4].max() - t0), 2), "entry_us_p50_max": [med(x[:, 0] - t0), round(float((x[:, 0] - t0).max()), 2)]}
    for i, ph in enumerate(phases):
        row[ph + " (us, median)"] = med(x[:, i + 1] - x[:, i])
    row["block_us_median"] = med(x[:, 4] - x[:, 0])
    print(json.dumps(row), flush=True)
