import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd.config import geometry_tiny
from socioreasoner_amd.engine import Engine
B = 48
e = Engine(geometry_tiny(), max_patches=256, max_prefill_tokens=64 * B, max_batch=B, max_ctx=128, max_new_tokens=24, kv_slots=2 * B)
e.load_synthetic_weights(seed=0)
e.rows_begin()
ids = [np.arange(5, 15, dtype=np.int64) + i for i in range(B)]
pos = [np.tile(np.arange(10), (3, 1)).astype(np.int64) for _ in range(B)]
e.admit(list(range(B)), ids, pos, [24] * B, None)
fin, cnt = e.rows_poll(); print("after admit fin", fin.tolist())
e.rows_step(2, [], 0)
fin, cnt = e.rows_poll(); print("after 2 steps fin", fin.tolist(), cnt.tolist()[:4])
e.rows_abort([3, 33, 47])
fin, cnt = e.rows_poll(); print("after abort fin", fin.tolist())
e.rows_step(2, [], 0)
fin, cnt = e.rows_poll(); print("after 2 more steps fin", fin.tolist(), "cnt", cnt.tolist())
# the scheduler-level sequence of tests/test_gpu_round4.py::test_rows_mode_above_32_rows_tiny_engine
from socioreasoner_amd.serving import ContinuousBatcher, Request
rng = np.random.default_rng(B)
ids2 = [rng.integers(0, 2000, int(rng.integers(5, 40))).astype(np.int64) for _ in range(B)]
pos2 = [np.tile(np.arange(len(x)), (3, 1)).astype(np.int64) for x in ids2]
cb = ContinuousBatcher(e, eos=[], pad_id=0, steps_per_poll=2)
for i in range(B):
    cb.submit(Request(ids=ids2[i], pos3=pos2[i], max_new=24, tag=i))
cb.pump(lambda r, t: print("done", r.tag))
print("active rows", sorted(cb.active.keys())[:5], len(cb.active))
victim = next(r for row, r in cb.active.items() if row >= 32)
vrow = next(row for row, r in cb.active.items() if r is victim)
print("victim row", vrow, "tag", victim.tag, "abort ->", cb.abort(lambda r: r is victim))
fin, cnt = e.rows_poll(); print("fin after abort", [i for i, f in enumerate(fin) if f], "cnt", cnt.tolist()[30:36])
cb.pump(lambda r, t: print("done", r.tag, r.aborted))
fin, cnt = e.rows_poll(); print("fin after pump", [i for i, f in enumerate(fin) if f])
print("victim still active:", victim.tag in [r.tag for r in cb.active.values()])
