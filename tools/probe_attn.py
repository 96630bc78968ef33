#!/usr/bin/env python3
"""PMC probe: two admissions (ViT + prefill) of 32 synthetic tiles on the 3B geometry, nothing else (tools/gpu_pmc_attn.sh)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from socioreasoner_amd import hostops, synthetic  # noqa: E402
from socioreasoner_amd.config import geometry_3b  # noqa: E402
from socioreasoner_amd.engine import Engine  # noqa: E402

B = 32
geom = geometry_3b()
e = Engine(geom, max_patches=1024 * B, max_prefill_tokens=448 * B, max_batch=B, max_ctx=512, max_new_tokens=4)
e.load_synthetic_weights(seed=0)
grid = (1, 32, 32)
imgs = [torch.from_numpy(synthetic.tile_pixels(i)).cuda() for i in range(B)]
ids, pos = [], []
for i in range(B):
    x = synthetic.tile_prompt(geom, i, grid)
    p, _ = hostops.get_rope_index(torch.from_numpy(x)[None], [grid], None)
    ids.append(x)
    pos.append(p[:, 0].numpy())
pix = torch.cat([e.patchify(im) for im in imgs], dim=0)
for _ in range(2):
    emb = e.vit_forward(pix, [grid] * B)
    e.prefill(ids, pos, emb)
torch.cuda.synchronize()
e.close()
