#!/usr/bin/env python3
"""SAM2 Hiera-L behind seg_infer, the measurement of bench.py's `sam2` object on its own: float32 mode with the split-bf16 GEMM (default),
float32 mode on the f32-input MFMA (SR_SAM_F32_SPLIT=0, round 4) and the bf16 mode.  One JSON object."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from socioreasoner_amd import lib, sam2, synthetic  # noqa: E402

dev = torch.device("cuda:0")
sg = sam2.Sam2Geometry()
simg = torch.from_numpy(synthetic.tile_pixels(0, 756, 756)).to(dev)
simgs = [torch.from_numpy(synthetic.tile_pixels(i, 756, 756)).to(dev) for i in range(8)]
sobj = [dict(point_coords=[[300 + 20 * k, 320]], point_labels=[1], box=[100 + 30 * k, 120, 420 + 30 * k, 600]) for k in range(4)]
ssd = sam2.synthetic_state_dict(sg)


def mode(dtype, split):
    os.environ["SR_SAM_F32_SPLIT"] = split
    lib.reload_switches()
    se = sam2.Sam2Engine(sg, str(dev), dtype=dtype)
    se.load_state_dict(ssd)
    sacc = torch.zeros(756, 756, dtype=torch.uint8, device=dev)

    def tiles8():
        se.set_images(simgs)
        for b_ in range(8):
            se.select(b_)
            se.predict_or_many(sacc, sobj)
    tiles8()
    t_ = {}
    for nm, fn, reps in (("set_images_8_ms", lambda: se.set_images(simgs), 3), ("predict_ms_4_objects_one_pass", lambda: se.predict_or_many(sacc, sobj), 20), ("tiles8_ms_4_objects", tiles8, 3)):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        t_[nm] = round((time.perf_counter() - t0) / reps * 1e3, 3)
    t_["tiles_per_s_4_objects_batched"] = round(8e3 / t_["tiles8_ms_4_objects"], 2)
    t_["mask_checksum"] = int(sacc.sum().item())
    del se
    torch.cuda.empty_cache()
    return t_


out = {"float32_split_bf16": mode(torch.float32, "1"), "float32_f32_mfma": mode(torch.float32, "0"), "bf16": mode(torch.bfloat16, "1")}
print(json.dumps(out))
