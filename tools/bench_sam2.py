"""GPU timing of the SAM2 path at Hiera-L: set_image(s) (756 x 756 tiles -> embeddings) and the mask decoder per object.  Synthetic weights."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import sam2, synthetic
g = sam2.Sam2Geometry()
e = sam2.Sam2Engine(g, dtype=__import__("torch").bfloat16)
e.load_state_dict(sam2.synthetic_state_dict(g))
img = torch.from_numpy(synthetic.tile_pixels(7, 756, 756)).cuda()
acc = torch.zeros(756, 756, dtype=torch.uint8, device="cuda")
imgs8 = [torch.from_numpy(synthetic.tile_pixels(7 + i, 756, 756)).cuda() for i in range(8)]
objs = [dict(point_coords=[[300 + k, 300]], point_labels=[1], box=[100 + 10 * k, 120, 500, 600]) for k in range(8)]
for nb in (1, 2, 4, 8):
    e.set_images(imgs8[:nb])
for n in (1, 2, 4, 8):
    e.predict_or_many(acc, objs[:n]); e.predict_or_many(acc, objs[:n])
torch.cuda.synchronize()
cases = [("set_image", lambda: e.set_image(img), 10, 1), ("set_images x2", lambda: e.set_images(imgs8[:2]), 5, 2), ("set_images x4", lambda: e.set_images(imgs8[:4]), 5, 4),
         ("set_images x8", lambda: e.set_images(imgs8), 5, 8)]
cases += [(f"decode {n} object(s) per pass", (lambda n: lambda: e.predict_or_many(acc, objs[:n]))(n), 20, n) for n in (1, 2, 4, 8)]
for name, fn, reps, units in cases:
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); ev0.record()
    for _ in range(reps): fn()
    ev1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:28s}: host issue {1e3 * (t1 - t0) / reps:7.3f} ms  wall {1e3 * (t2 - t0) / reps:7.3f} ms  gpu {ev0.elapsed_time(ev1) / reps:7.3f} ms  = {ev0.elapsed_time(ev1) / reps / units:6.3f} ms per unit")
