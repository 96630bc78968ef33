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
