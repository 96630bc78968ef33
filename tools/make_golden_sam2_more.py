#!/usr/bin/env python3
"""MORE golden fixtures of the SAM2 image path at Hiera-L ("exactness holds for one image, three prompts"):
HF ``Sam2Model`` in float32 -- the reference's precision -- on two further synthetic 756 x 756 images and a SECOND set of synthetic weights
(seed 1), with multi-point prompts: three clicks of mixed labels without a box, a box with three clicks, a small box at the image edge, a
negative click inside a box.  Same construction as tools/make_golden_sam2.py (whose hf_model / prompt preparation it imports); float32 only,
and per prompt only what the mask test needs: the three IoU scores, the BEST mask's low-resolution logits and its packed 756 x 756 mask.
Runs ONLY in the build container; stores inputs / outputs only -> tests/golden/sam2_more.npz.
Usage: python tools/make_golden_sam2_more.py"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import sam2_ref as S  # noqa: E402
from socioreasoner_amd import synthetic  # noqa: E402
from make_golden_sam2 import hf_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sam2_more.npz")
CASES = [   # (image seed, weight seed, prompts in the 756 x 756 frame)
    (11, 0, [dict(box=None, pts=[[200, 240], [420, 300], [610, 520]], labels=[1, 1, 0]),
             dict(box=[150, 200, 640, 700], pts=[[300, 350], [500, 600], [180, 220]], labels=[1, 1, 0]),
             dict(box=[0, 0, 90, 130], pts=None, labels=None)]),
    (23, 1, [dict(box=[260, 90, 730, 480], pts=[[500, 250]], labels=[0]),
             dict(box=None, pts=[[100, 650], [130, 700], [60, 610], [700, 60]], labels=[1, 1, 1, 0]),
             dict(box=[400, 400, 755, 755], pts=[[600, 600], [450, 700]], labels=[1, 1])]),
]


def main():
    torch.manual_seed(0)
    g = S.geometry_large()
    out = {"n_cases": np.array([len(CASES)])}
    for ci, (iseed, wseed, prompts) in enumerate(CASES):
        t0 = time.time()
        W = S.synthetic_weights(g, seed=wseed)
        model = hf_model(g, W, torch.float32)
        img = synthetic.tile_pixels(iseed, 756, 756)
        px = S.preprocess(img, g.image_size, torch.float32)
        out[f"c{ci}_img_seed"], out[f"c{ci}_weight_seed"], out[f"c{ci}_n_prompts"] = np.array([iseed]), np.array([wseed]), np.array([len(prompts)])
        with torch.no_grad():
            emb = model.get_image_embeddings(px)
            for p, pr in enumerate(prompts):
                c, l = S.prompt_points(pr["box"], pr["pts"], pr["labels"], (756, 756), g.image_size)
                o = model(image_embeddings=emb, input_points=c[None, None], input_labels=l[None, None].int(), multimask_output=True)
                low, iou = o.pred_masks[0, 0], o.iou_scores[0, 0]
                best, masks, up = S.postprocess(low, iou, (756, 756))
                b = int(torch.argmax(iou))
                k = f"c{ci}_p{p}"
                out[k + "_box"] = np.array(pr["box"] if pr["box"] is not None else [], dtype=np.float32)
                out[k + "_pts"] = np.array(pr["pts"] if pr["pts"] is not None else [], dtype=np.float32).reshape(-1, 2)
                out[k + "_labels"] = np.array(pr["labels"] if pr["labels"] is not None else [], dtype=np.int64)
                out[k + "_iou"] = iou.numpy().astype(np.float32)
                out[k + "_best"] = np.array([b])
                out[k + "_low_best"] = low[b].numpy().astype(np.float32)
                out[k + "_mask_bits"] = np.packbits(best.numpy())
                out[k + "_band"] = np.array([int((up[b].abs() < 1e-3).sum())])
                print(f"case {ci} p{p}: iou {iou.numpy()} best {b} mask area {int(best.sum())} pixels inside the 1e-3 band {int(out[k + '_band'][0])}", flush=True)
        print(f"case {ci}: {time.time() - t0:.1f}s", flush=True)
        del model, W
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
