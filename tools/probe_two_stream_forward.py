"""Would running the ViT + prefill of a 32-tile admission as TWO independent 16-tile halves on two streams fill the GEMMs' partial last rounds
and the small launches' latency?  Two engines (B = 16 each) on two torch streams against one engine (B = 32), same tiles, wall time per 32 tiles."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import hostops, synthetic
from socioreasoner_amd.config import geometry_3b
from socioreasoner_amd.engine import Engine

geom = geometry_3b()
GRID, NP, S = (1, 32, 32), 1024, 448
dev = torch.device("cuda:0")


def make(B):
    e = Engine(geom, max_patches=NP * B, max_prefill_tokens=S * B, max_batch=B, max_ctx=640, max_new_tokens=128, device=str(dev))
    e.load_synthetic_weights(seed=0)
    return e


def inputs(lo, n):
    imgs = [torch.from_numpy(synthetic.tile_pixels(i, 448, 448)).to(dev) for i in range(lo, lo + n)]
    ids = [synthetic.tile_prompt(geom, i, GRID) for i in range(lo, lo + n)]
    pos3 = [hostops.get_rope_index(torch.from_numpy(x)[None], [GRID], None, image_token_id=geom.image_token_id, vision_start_token_id=geom.vision_start_token_id)[0][:, 0].numpy() for x in ids]
    return imgs, ids, pos3


def forward(e, imgs, ids, pos3):
    pix = torch.cat([e.patchify(im) for im in imgs], dim=0)
    emb = e.vit_forward(pix, [GRID] * len(imgs))
    e.prefill(ids, pos3, emb)


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


e32 = make(32)
in32 = inputs(0, 32)
t_one = timed(lambda: forward(e32, *in32))
del e32
torch.cuda.empty_cache()
ea, eb = make(16), make(16)
ina, inb = inputs(0, 16), inputs(16, 16)
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def two():
    with torch.cuda.stream(sa):
        forward(ea, *ina)
    with torch.cuda.stream(sb):
        forward(eb, *inb)


def two_serial():
    forward(ea, *ina)
    forward(eb, *inb)


t_two = timed(two)
t_ser = timed(two_serial)
print(f"one engine, 32 tiles: {t_one:.2f} ms   two engines x 16 tiles on two streams: {t_two:.2f} ms   the same two, one stream: {t_ser:.2f} ms")
