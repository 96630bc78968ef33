"""Does a decode step pay for the engine's context CAPACITY or for the rows' actual contexts?  The same 32 prompts of 448 tokens, 64 greedy tokens,
on engines built with max_ctx = 640 (the bench), 2176 and 6144 (examples/infer/rlvr_megatron.yaml: prompt_length 4096 + response_length 2048)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import hostops, synthetic
from socioreasoner_amd.config import geometry_3b
from socioreasoner_amd.engine import Engine

geom = geometry_3b()
B, G = 32, 64
grid = (1, 32, 32)
imgs = [torch.from_numpy(synthetic.tile_pixels(i)).cuda() for i in range(B)]
ids = [synthetic.tile_prompt(geom, i, grid) for i in range(B)]
pos = [hostops.get_rope_index(torch.from_numpy(x)[None], [grid], None)[0][:, 0].numpy() for x in ids]
ref = None
for ctx in (640, 2176, 6144):
    e = Engine(geom, max_patches=1024 * B, max_prefill_tokens=448 * B, max_batch=B, max_ctx=ctx, max_new_tokens=G)
    e.load_synthetic_weights(seed=0)
    def forward():
        pix = torch.cat([e.patchify(im) for im in imgs], dim=0)
        emb = e.vit_forward(pix, [grid] * B)
        e.prefill(ids, pos, emb)
    forward(); toks = e.decode(G).cpu()
    forward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e.decode(G); e1.record(); torch.cuda.synchronize()
    if ref is None:
        ref = toks
    print(json.dumps({"max_ctx": ctx, "decode_ms_per_step": round(e0.elapsed_time(e1) / (G - 1), 3), "tokens_equal_to_max_ctx_640": bool(torch.equal(toks, ref)),
                      "workspace_GB": round(e.workspace_bytes / 2 ** 30, 2) if hasattr(e, "workspace_bytes") else None}), flush=True)
    e.close()
