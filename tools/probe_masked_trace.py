"""Kernel trace of the same 63 decode steps of a 32-row engine on ONE chosen stream -- `plain` (ordinary), `m8` (CU mask with every CU enabled), `m7` .. `m3` (7 .. 3 of the 8 CUs
of every shader engine) -- so that rocprofv3's per-kernel durations and the gaps between them can be compared stream by stream:
    rocprofv3 --kernel-trace --stats -d out -o t -- python tools/probe_masked_trace.py m5
Is a decode step on the scheduler's 160-CU stream slower because its kernels are longer, or because the launches are further apart?  (round 6)
On the m7 .. m3 streams the engine gets the scheduler's CU hint (sr_rows_set_cus) unless PROBE_NO_HINT=1."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import hostops, synthetic, streams
from socioreasoner_amd.config import geometry_3b
from socioreasoner_amd.engine import Engine

which = sys.argv[1] if len(sys.argv) > 1 else "plain"
geom = geometry_3b()
B, G, grid = 32, 64, (1, 32, 32)
e = Engine(geom, max_patches=1024 * B, max_prefill_tokens=448 * B, max_batch=B, max_ctx=640, max_new_tokens=G)
e.load_synthetic_weights(seed=0)
imgs = [torch.from_numpy(synthetic.tile_pixels(i)).cuda() for i in range(B)]
ids = [synthetic.tile_prompt(geom, i, grid) for i in range(B)]
pos = [hostops.get_rope_index(torch.from_numpy(x)[None], [grid], None)[0][:, 0].numpy() for x in ids]
plain = torch.cuda.Stream()
s = plain if which == "plain" else streams.masked_stream("cuda:0", 0, int(which[1:]))
with torch.cuda.stream(plain):
    pix = torch.cat([e.patchify(im) for im in imgs], dim=0)
    e.prefill(ids, pos, e.vit_forward(pix, [grid] * B))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(s):
    if which[0] == "m" and which != "m8" and os.environ.get("PROBE_NO_HINT") != "1":
        e.rows_set_cus(int(which[1:]) * 32)          # the scheduler's hint: the step's x-stationary form (sr_rows_set_cus; PROBE_NO_HINT=1: the streaming kernels)
    a.record()
    e.decode(G)
    b.record()
torch.cuda.synchronize()
print(which, "decode ms per step", round(a.elapsed_time(b) / (G - 1), 3))
e.close()
