"""Where does a decode GEMV launch's time go?  The four weight-streaming launches of one LM layer at 32 batch rows, on a build of gemv.hip
with -DSR_GEMV_TIMING (socioreasoner_amd/libsocior_timing.so; never the product library): wave 0 of every block records the 100 MHz clock
at entry, when its ring is filled and the k loop starts, after the k loop and at its end.  Weights rotate through R copies (no cache reuse);
the stamps of the last launch are read.  Per launch: when the blocks START (launch ramp), how long a block runs and how that splits, when
they END (tail), against the launch's span and the time the bytes would take at 6.3 TB/s."""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = C.CDLL(os.path.join(ROOT, "socioreasoner_amd", os.environ.get("SR_TIMING_LIB", "libsocior_timing.so")))
vp, ci = C.c_void_p, C.c_int
L.sr_op_gemv.argtypes = [vp, ci, vp, ci, ci, ci, vp, ci, ci, vp]
L.sr_op_gemv_fused.argtypes = [vp, ci, vp, ci, ci, ci, vp, ci, ci, vp, vp, C.c_float, vp, ci, vp, vp, vp, vp]
L.sr_dbg_gemv_times.argtypes = [vp, ci]
P = lambda t: vp(t.data_ptr()) if t is not None else None
s = vp(torch.cuda.current_stream().cuda_stream)
B = int(os.environ.get("PB", 32))
H, QN, I = 2048, 2560, 11008
R = 12
XT, OT, WT = 0x800, 0x1000, 0x100
x = torch.randn(32, I, device="cuda").to(torch.bfloat16)
bias = torch.zeros(QN, device="cuda").to(torch.bfloat16)
mk = lambda n, k: (torch.randn(R, n, k, device="cuda") * 0.02).to(torch.bfloat16)
cases = [
    ("qkv (bias)", mk(QN, H), QN * H * 2, torch.zeros(B, QN, dtype=torch.bfloat16, device="cuda"),
     lambda w, o: L.sr_op_gemv_fused(P(x), H, P(w), B, QN, H, P(o), QN, 3 | WT | XT, P(bias), None, 0.0, None, 0, None, None, None, s)),
    ("o (resid)", mk(H, H), H * H * 2, torch.zeros(B, H, dtype=torch.bfloat16, device="cuda"),
     lambda w, o: L.sr_op_gemv_fused(P(x), H, P(w), B, H, H, P(o), H, 4 | WT | XT, None, None, 0.0, None, 0, None, None, None, s)),
    ("gate/up (swiglu)", mk(2 * I, H), 2 * I * H * 2, torch.zeros(B, I, dtype=torch.bfloat16, device="cuda"),
     lambda w, o: L.sr_op_gemv_fused(P(x), H, P(w), B, 2 * I, H, P(o), I, 1 | WT | XT | OT, None, None, 0.0, None, 0, None, None, None, s)),
    ("down (partial, ksplit 4)", mk(H, I), H * I * 2, torch.zeros(4, B, H, device="cuda"),
     lambda w, o: L.sr_op_gemv(P(x), I, P(w), B, H, I, P(o), 4, 0 | WT | XT, s)),
]
V = 151936
nblk = 4096
av = torch.zeros(32, nblk, device="cuda"); ai = torch.zeros(32, nblk, dtype=torch.int32, device="cuda")
cases.append(("lm head (f32 + argmax)", (torch.randn(2, V, H, device="cuda") * 0.02).to(torch.bfloat16), V * H * 2, torch.zeros(B, V, device="cuda"),
              lambda w, o: L.sr_op_gemv_fused(P(x), H, P(w), B, V, H, P(o), V, 2 | WT | XT, None, None, 0.0, None, 0, None, P(av), P(ai), s)))
for name, W, nbytes, out, fn in cases:
    R = W.shape[0]
    for it in range(3):
        for r in range(R):
            assert fn(W[r], out) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(5):
        for r in range(R):
            fn(W[r], out)
    e1.record(); torch.cuda.synchronize()
    us_launch = e0.elapsed_time(e1) * 1e3 / (5 * R)
    t = np.zeros((16384, 4), np.int64)
    assert L.sr_dbg_gemv_times(t.ctypes.data, 16384) == 0
    # blocks of the LAST launch: stamps newer than ... take all rows whose t0 is within 100 us of the newest stamp
    newest = t[:, 3].max()
    live = (t[:, 0] > newest - int(us_launch * 130)) & (t[:, 3] >= t[:, 0])
    tt = t[live].astype(np.float64) * 0.01
    t0 = tt[:, 0].min()
    q = lambda v, p_: round(float(np.percentile(v, p_)), 2)
    row = {"launch": name, "rows": B, "blocks": int(live.sum()), "MB": round(nbytes / 1e6, 2), "us_per_launch_back_to_back": round(us_launch, 2),
           "us_at_6.3TBs": round(nbytes / 6.3e6, 2), "span_first_entry_to_last_exit_us": round(float(tt[:, 3].max() - t0), 2),
           "entry_us_p50_p90_max": [q(tt[:, 0] - t0, 50), q(tt[:, 0] - t0, 90), q(tt[:, 0] - t0, 100)],
           "ring_filled_after_entry_us_p50": q(tt[:, 1] - tt[:, 0], 50),
           "k_loop_us_p50_p90_max": [q(tt[:, 2] - tt[:, 1], 50), q(tt[:, 2] - tt[:, 1], 90), q(tt[:, 2] - tt[:, 1], 100)],
           "reduce_epilogue_us_p50": q(tt[:, 3] - tt[:, 2], 50),
           "block_us_p50_p90_max": [q(tt[:, 3] - tt[:, 0], 50), q(tt[:, 3] - tt[:, 0], 90), q(tt[:, 3] - tt[:, 0], 100)],
           "exit_us_p10_p50_p90_max": [q(tt[:, 3] - t0, 10), q(tt[:, 3] - t0, 50), q(tt[:, 3] - t0, 90), q(tt[:, 3] - t0, 100)]}
    print(json.dumps(row), flush=True)
