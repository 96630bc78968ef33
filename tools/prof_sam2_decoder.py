"""rocprofv3 target: the SAM2 mask decoder, 4 objects of one tile per pass, replayed from its captured graph (tools/gpu_r3_sam_dec_prof.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import sam2, synthetic
g = sam2.Sam2Geometry()
e = sam2.Sam2Engine(g, dtype=__import__("torch").bfloat16)
e.load_state_dict(sam2.synthetic_state_dict(g))
e.set_image(torch.from_numpy(synthetic.tile_pixels(7, 756, 756)).cuda())
acc = torch.zeros(756, 756, dtype=torch.uint8, device="cuda")
objs = [dict(point_coords=[[300 + 20 * k, 320]], point_labels=[1], box=[100 + 30 * k, 120, 420 + 30 * k, 600]) for k in range(4)]
for _ in range(60):
    e.predict_or_many(acc, objs)
torch.cuda.synchronize()
