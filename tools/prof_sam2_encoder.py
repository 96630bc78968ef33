"""rocprofv3 target: the SAM2 Hiera-L image encoder over 8 tiles per pass (tools/gpu_r3_sam_prof.sh); `f32` as the first argument = the
float32 mode (the reference's precision, round 4)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import sam2, synthetic
g = sam2.Sam2Geometry()
e = sam2.Sam2Engine(g, dtype=torch.float32 if sys.argv[1:2] == ["f32"] else torch.bfloat16)
e.load_state_dict(sam2.synthetic_state_dict(g))
imgs = [torch.from_numpy(synthetic.tile_pixels(7 + i, 756, 756)).cuda() for i in range(8)]
for _ in range(4):
    e.set_images(imgs)
torch.cuda.synchronize()
