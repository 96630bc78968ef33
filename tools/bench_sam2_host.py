"""SAM2 behind seg_infer from HOST images (PIL-sized uint8 arrays, as the pipeline hands them over): 32 tiles with 4 objects each through
Sam2Predictor.segment_batch -- uploads, encoder (8 per pass), decoder, union -- wall time per tile; second call = the embedding cache."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import sam2, synthetic
g = sam2.Sam2Geometry()
e = sam2.Sam2Engine(g, dtype=__import__("torch").bfloat16)
e.load_state_dict(sam2.synthetic_state_dict(g))
pr = sam2.Sam2Predictor(e)
N = 32
imgs = [synthetic.tile_pixels(100 + i, 756, 756) for i in range(N)]
objs = [[dict(point_coords=[[300 + 20 * k, 320]], point_labels=[1], box=[100 + 30 * k, 120, 420 + 30 * k, 600]) for k in range(4)] for _ in range(N)]
warm = [synthetic.tile_pixels(900 + i, 756, 756) for i in range(8)]
pr.segment_batch(warm, objs[:8]); torch.cuda.synchronize()
for name in ("first pass (encode)", "second pass (cached embeddings)"):
    t0 = time.perf_counter()
    out = pr.segment_batch(imgs, objs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:32s}: {1e3 * dt / N:6.2f} ms per tile  ({N / dt:6.1f} tiles/s)  stats {pr.stats}")
