"""The five weight-streaming launches of a 32-row decode layer (+ LM head), each timed back to back over distinct weight copies, on the ORDINARY stream and on
CU-masked streams (7 / 5 of the 8 CUs of every shader engine: the scheduler's decode stream is 5 / 8 while an admission is staged) -- and the same with a library
built with -DSR_EXP_NOX (activation loads pinned to chunk 0: no L2 traffic for x, wrong results): is a launch on 160 CUs slower because a CU cannot pull more
HBM-missing bytes, or because half of what it pulls is x from L2?  (round 6; SR_LIB_PATH selects the library; PROBE_HINT=1: what the x-stationary kernels make of the masked streams)"""
import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib, streams
if os.environ.get("SR_LIB_PATH"):
    lib.LIB_PATH = os.environ["SR_LIB_PATH"]
L = lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
B, H, QN, I, V, R = 32, 2048, 2560, 11008, 151936, 12
XT, OT, TL = 0x800, 0x1000, 0x100
x = torch.randn(B, I, device="cuda").to(torch.bfloat16)
bias = torch.zeros(QN, device="cuda").to(torch.bfloat16)
mk = lambda n, k, r=R: (torch.randn(r, n, k, device="cuda") * 0.02).to(torch.bfloat16)
nb = L.sr_op_gemv_f32_blocks(V, B, H, 0)
av, ai = torch.zeros(B, nb, device="cuda"), torch.zeros(B, nb, dtype=torch.int32, device="cuda")
eps = C.c_float(1e-6)
cases = [
    ("q/k/v", mk(QN, H), lambda w, o, s: L.sr_op_gemv_fused(P(x), I, P(w), B, QN, H, P(o), QN, 3 | TL | XT, P(bias), None, eps, None, 0, None, None, None, s), torch.zeros(B, QN, dtype=torch.bfloat16, device="cuda")),
    ("o_proj", mk(H, H), lambda w, o, s: L.sr_op_gemv_fused(P(x), I, P(w), B, H, H, P(o), H, 4 | TL | XT, None, None, eps, None, 0, None, None, None, s), torch.zeros(B, H, dtype=torch.bfloat16, device="cuda")),
    ("gate/up", mk(2 * I, H), lambda w, o, s: L.sr_op_gemv_fused(P(x), I, P(w), B, 2 * I, H, P(o), I, 1 | TL | XT | OT, None, None, eps, None, 0, None, None, None, s), torch.zeros(B, I, dtype=torch.bfloat16, device="cuda")),
    ("down (4 slabs)", mk(H, I), lambda w, o, s: L.sr_op_gemv(P(x), I, P(w), B, H, I, P(o), 4, 0 | TL | XT, s), torch.zeros(4, B, H, device="cuda")),
    ("LM head", mk(V, H, 2), lambda w, o, s: L.sr_op_gemv_fused(P(x), I, P(w), B, V, H, P(o), V, 2 | TL | XT, None, None, eps, None, 0, None, P(av), P(ai), s), torch.zeros(B, V, device="cuda")),
]
strs = {"ordinary": (torch.cuda.Stream(), 0), "7 of 8 CUs": (streams.masked_stream("cuda:0", 0, 7), 224), "5 of 8 CUs": (streams.masked_stream("cuda:0", 0, 5), 160)}
HINT = os.environ.get("PROBE_HINT", "0") == "1"      # PROBE_HINT=1: the masked streams get the scheduler's CU hint (sr_op_gemv_set_cus: x-stationary gate/up and down)
out = {}
for name, W, fn, o in cases:
    for sn, (st, cus) in strs.items():
        sp = C.c_void_p(st.cuda_stream)
        with torch.cuda.stream(st):
            assert L.sr_op_gemv_set_cus(cus if HINT else 0, sp) == 0
            for r in range(W.shape[0]):
                assert fn(W[r], o, sp) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                for r in range(W.shape[0]):
                    fn(W[r], o, sp)
            e1.record()
        torch.cuda.synchronize()
        out.setdefault(name, {})[sn] = round(e0.elapsed_time(e1) * 1e3 / (5 * W.shape[0]), 2)
print(json.dumps({"library": os.path.basename(lib.LIB_PATH), "cu_hint_on_masked_streams": HINT, "us_per_launch": out}))
