#!/usr/bin/env python3
"""Soak test of the overlapped-admission scheduler on the 3B geometry: random mixes of image / text-only requests with ragged prompt
lengths and budgets through ContinuousBatcher(overlap=True) must give exactly the tokens of the one-stream scheduler, round after
round (races between the admission stream and the decode stream would show up as differences or hangs).  Usage: soak_overlap.py [rounds]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from socioreasoner_amd import hostops, synthetic  # noqa: E402
from socioreasoner_amd.config import geometry_3b  # noqa: E402
from socioreasoner_amd.engine import Engine  # noqa: E402
from socioreasoner_amd.serving import ContinuousBatcher, Request  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
geom = geometry_3b()
B, G = 32, 40
e = Engine(geom, max_patches=1024 * B, max_prefill_tokens=448 * B, max_batch=B, max_ctx=512, max_new_tokens=G, kv_slots=56)
e.load_synthetic_weights(seed=0)
grid = (1, 32, 32)
imgs = [torch.from_numpy(synthetic.tile_pixels(i)).cuda() for i in range(8)]
bad = 0
for rnd in range(rounds):
    rng = np.random.default_rng(100 + rnd)
    n = int(rng.integers(40, 90))
    reqs_spec = []
    for i in range(n):
        if rng.random() < 0.35:
            x = rng.integers(1000, 60000, size=int(rng.integers(8, 120))).astype(np.int64)
            p, _ = hostops.get_rope_index(torch.from_numpy(x)[None], None, None)
            reqs_spec.append((x, p[:, 0].numpy(), int(rng.integers(1, 12)), None))
        else:
            x = synthetic.tile_prompt(geom, int(rng.integers(0, 1000)), grid)[: 448 - int(rng.integers(0, 40))]
            p, _ = hostops.get_rope_index(torch.from_numpy(x)[None], [grid], None)
            reqs_spec.append((x, p[:, 0].numpy(), int(rng.integers(2, G + 1)), int(rng.integers(0, 8))))
    mk = lambda: [Request(ids=x, pos3=p, max_new=m, images=[imgs[k]] if k is not None else [], grids=[grid] if k is not None else []) for x, p, m, k in reqs_spec]
    t0 = time.time()
    a = ContinuousBatcher(e, eos=[], pad_id=0, steps_per_poll=int(rng.choice([1, 2, 4, 8])), overlap=True).run(mk())
    t1 = time.time()
    b = ContinuousBatcher(e, eos=[], pad_id=0, steps_per_poll=4).run(mk())
    ok = a == b
    bad += not ok
    print(f"round {rnd}: {n} requests, overlap {t1 - t0:.2f} s, one-stream {time.time() - t1:.2f} s, equal={ok}", flush=True)
e.close()
print("SOAK", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
