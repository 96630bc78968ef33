#!/usr/bin/env python3
"""Pins what CAN be pinned of the SAM2 predictor's pre / post-processing contract (missing #3).

The reference calls the sam2 package's ``SAM2ImagePredictor.set_image / predict`` (/root/reference/roll/distributed/strategy/
seg_strategy.py:47-60); that package is not installed here, so ``oracle/sam2_ref.py`` restates its prompt preparation and mask
post-processing.  HF transformers ships an independent implementation of the same contract (``Sam2Processor`` +
``Sam2ImageProcessor`` + ``Sam2PromptEncoder``); this script runs THOSE on the tiny-geometry fixture model of tools/make_golden_sam2.py
and stores inputs / outputs only (tests/golden/sam2_contract.npz):

  * prompt scaling          ``Sam2Processor.__call__(original_sizes=..., input_points, input_labels, input_boxes)`` -- executed as shipped
                            (the processor's prompt path does not touch its image processor);
  * box handling            HF's NATIVE box path (``input_boxes`` -> ``_embed_boxes``: corners + 0.5, point_embed[2] / [3], the padding
                            point) through ``Sam2Model`` in float32, next to the box-as-two-labelled-points form the sam2 package's predictor
                            (and this repo) uses -- the low-resolution logits / IoUs of both must agree to float32 round-off;
  * mask post-processing    ``Sam2ImageProcessor.post_process_masks`` (bilinear, align_corners=False, > 0.0).  Its module imports torchvision
                            (absent), so the method is taken from the installed source file by AST and executed on a stub ``self`` --
                            the same technique tools/make_golden.py uses for the reference's own pure functions.

NOT pinnable offline (said in DESIGN.md): the image resize + normalisation of ``SAM2Transforms`` (torchvision) and the
``sam2_hiera_large.pt`` parameter-name table.  Runs ONLY in the build container.  Usage: python tools/make_golden_sam2_contract.py
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import sam2_ref as S  # noqa: E402
from socioreasoner_amd import synthetic  # noqa: E402
from make_golden_sam2 import PROMPTS, hf_model  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sam2_contract.npz")


def hf_post_process_masks():
    """``Sam2ImageProcessor.post_process_masks`` from the installed transformers, without importing its module (torchvision)."""
    import transformers
    import torch.nn.functional as F
    path = os.path.join(os.path.dirname(transformers.__file__), "models", "sam2", "image_processing_sam2.py")
    tree = ast.parse(open(path).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "Sam2ImageProcessor")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "post_process_masks")
    ns = {"torch": torch, "np": np, "F": F}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns["post_process_masks"]


def main():
    from transformers.models.sam2.processing_sam2 import Sam2Processor
    tag = "tiny"
    g = S.geometry_tiny()
    W = S.synthetic_weights(g)
    hw = 189
    img = synthetic.tile_pixels(7, hw, hw)
    model = hf_model(g, W, torch.float32)
    proc = Sam2Processor.__new__(Sam2Processor)          # (its __init__ wants a real image processor; the prompt path uses these two fields only)
    proc.target_size, proc.point_pad_value = g.image_size, -10
    post = hf_post_process_masks()
    out = {"hw": np.array([hw]), "image_size": np.array([g.image_size])}
    with torch.no_grad():
        emb = model.get_image_embeddings(S.preprocess(img, g.image_size, torch.float32))
        for p, pr in enumerate(PROMPTS[tag]):
            kw = {}
            if pr["pts"] is not None:
                kw["input_points"], kw["input_labels"] = [[pr["pts"]]], [[pr["labels"]]]
            if pr["box"] is not None:
                kw["input_boxes"] = [[pr["box"]]]
            enc = proc(original_sizes=[[hw, hw]], return_tensors="pt", **kw)
            out[f"p{p}_box"] = np.array(pr["box"] if pr["box"] is not None else [], dtype=np.float32)
            out[f"p{p}_pts"] = np.array(pr["pts"] if pr["pts"] is not None else [], dtype=np.float32).reshape(-1, 2)
            out[f"p{p}_labels"] = np.array(pr["labels"] if pr["labels"] is not None else [], dtype=np.int64)
            out[f"p{p}_hf_points"] = enc["input_points"][0, 0].numpy() if "input_points" in enc else np.zeros((0, 2), np.float32)
            out[f"p{p}_hf_boxes"] = enc["input_boxes"][0, 0].numpy() if "input_boxes" in enc else np.zeros((0,), np.float32)
            o = model(image_embeddings=emb, input_points=enc.get("input_points"), input_labels=enc["input_labels"].int() if "input_labels" in enc else None,
                      input_boxes=enc.get("input_boxes"), multimask_output=True)
            low, iou = o.pred_masks[0, 0], o.iou_scores[0, 0]
            out[f"p{p}_low_native"] = low.numpy().astype(np.float32)
            out[f"p{p}_iou_native"] = iou.numpy().astype(np.float32)
            masks = post(None, [o.pred_masks[0]], [[hw, hw]])[0][0]                      # [3, hw, hw] bool
            out[f"p{p}_masks_bits"] = np.packbits(masks.numpy().astype(np.uint8))
            # the form this repo (and the sam2 package's predictor) uses: box corners as points labelled 2 / 3 in FRONT of the clicks
            c, l = S.prompt_points(pr["box"], pr["pts"], pr["labels"], (hw, hw), g.image_size)
            o2 = model(image_embeddings=emb, input_points=c[None, None], input_labels=l[None, None].int(), multimask_output=True)
            d = float((o2.pred_masks[0, 0] - low).abs().max())
            print(f"p{p}: native-box vs box-as-points low-res logits max diff {d:.2e}; |logit|max {float(low.abs().max()):.2f}; iou {iou.numpy()}")
            assert d < 1e-4
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
