#!/usr/bin/env python3
"""Where does the rows-mode (continuous batching) decode step lose its 6 % against the static decode loop?  Same engine, same 32 prompts:
  A  static: prefill + sr_decode(128) (one graph replay per step, no host sync in between)
  B  rows mode, ONE chunk of 127 steps (no poll in between)
  C  rows mode, chunks of 16 steps with a poll (stream sync) after each -- what the scheduler does
  D  as C, on a second torch stream (the scheduler's decode_full stream)
GPU time by events on the launch stream + host wall time."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import hostops, synthetic
from socioreasoner_amd.config import geometry_3b
from socioreasoner_amd.engine import Engine

B, NEW, GRID = 32, 128, (1, 32, 32)
geom = geometry_3b()
for kv in (0, 64):
    eng = Engine(geom, max_patches=1024 * B, max_prefill_tokens=448 * B, max_batch=B, max_ctx=640, max_new_tokens=192, kv_slots=kv)
    eng.load_synthetic_weights(seed=0)
    dev = eng.device
    imgs = [torch.from_numpy(synthetic.tile_pixels(i)).to(dev) for i in range(B)]
    ids = [synthetic.tile_prompt(geom, i, GRID) for i in range(B)]
    pos3 = [hostops.get_rope_index(torch.from_numpy(x)[None], [GRID], None, image_token_id=geom.image_token_id, vision_start_token_id=geom.vision_start_token_id)[0][:, 0].numpy() for x in ids]
    pix = torch.cat([eng.patchify(im) for im in imgs], dim=0)
    emb = eng.vit_forward(pix, [GRID] * B)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def timed(fn, stream=None):
        s = stream or torch.cuda.current_stream(dev)
        torch.cuda.synchronize(dev)
        a, b = ev(), ev()
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            a.record(s)
            fn()
            b.record(s)
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b), (time.perf_counter() - t0) * 1e3

    for rep in range(2):
        eng.prefill(ids, pos3, emb)
        g, w = timed(lambda: eng.decode(NEW))
        print(f"kv_slots={kv} A static decode: {g / (NEW - 1):.4f} ms/step gpu, {w / (NEW - 1):.4f} wall", flush=True)
    for name, chunk, side in (("B one chunk", NEW - 1, False), ("C chunks of 16 + poll", 16, False), ("D chunks of 16 + poll, side stream", 16, True), ("E chunks of 16, no poll", 16, None)):
        for rep in range(2):
            eng.rows_begin()
            eng.admit(list(range(B)), ids, pos3, [NEW] * B, emb)
            side_s = torch.cuda.Stream(dev) if side else None
            if side_s is not None:
                side_s.wait_stream(torch.cuda.current_stream(dev))

            def run():
                left = NEW - 1
                while left > 0:
                    n = min(chunk, left)
                    eng.rows_step(n, [], 0)
                    if side is not None and chunk < NEW - 1:
                        eng.rows_poll()
                    left -= n
            g, w = timed(run, side_s)
            print(f"kv_slots={kv} {name}: {g / (NEW - 1):.4f} ms/step gpu, {w / (NEW - 1):.4f} wall", flush=True)
    eng.close()
    del eng
    torch.cuda.empty_cache()
