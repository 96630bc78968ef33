import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import sam2, synthetic
g = sam2.Sam2Geometry()
e = sam2.Sam2Engine(g, dtype=__import__("torch").bfloat16)
e.load_state_dict(sam2.synthetic_state_dict(g))
img = torch.from_numpy(synthetic.tile_pixels(7, 756, 756)).cuda()
acc = torch.zeros(756, 756, dtype=torch.uint8, device="cuda")
e.set_image(img)
for _ in range(3): e.predict_or(acc, [[300, 300]], [1], [100, 120, 500, 600])
torch.cuda.synchronize()
print("threads", torch.get_num_threads())
pr = cProfile.Profile(); pr.enable()
for _ in range(20): e.predict_or(acc, [[300, 300]], [1], [100, 120, 500, 600])
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
