#!/usr/bin/env python3
"""Golden fixtures of the SAM2 image path (SURVEY.md "next" row N1): HF ``transformers.models.sam2.Sam2Model`` (transformers 5.15.0,
eager attention) at the Hiera-L geometry of ``facebook/sam2-hiera-large`` -- the model the reference's seg_infer role loads
(/root/reference/roll/models/model_providers.py:540-541, examples/infer/rlvr_megatron.yaml:109-118) -- and at a tiny geometry, on
synthetic weights (oracle/sam2_ref.py: synthetic_weights, the repo's counter-based generator) and synthetic 756 x 756 images, in
float32 (the reference's numerics: SAM2 runs without autocast there) and in bfloat16 (the calibration of what a bf16 device path
can be held to).  The sam2 package itself is not installed; its predictor's pre / post-processing is applied as restated in
oracle/sam2_ref.py (preprocess / prompt_points / postprocess).  Runs ONLY in the build container; stores inputs / outputs only.

tests/golden/sam2.npz, per geometry tag (tiny, large) and prompt p (0: box, 1: box + two clicks, 2: one click):
  {tag}_img_seed / hw, {tag}_p{p}_box / pts / labels
  float32 run : {tag}_p{p}_low [3, m, m] low-resolution mask logits, _iou [3], _best (arg-max), _mask_bits (packed 756 x 756 mask of the best)
                {tag}_feat{0,1,2}_f32 strided samples of conv_s0(fpn0), conv_s1(fpn1), image embedding; {tag}_stage{0..3}_f32 of the stage outputs
  bfloat16 run: {tag}_p{p}_low_bf16, _iou_bf16, {tag}_feat*_bf16, {tag}_stage*_bf16 (same elements)
Usage: python tools/make_golden_sam2.py
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sam2_ref as S  # noqa: E402
from socioreasoner_amd import synthetic  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "sam2.npz")
STRIDE = {"tiny": 7, "large": 97}
PROMPTS = {  # in the 756 x 756 (large) / 189 x 189 (tiny) frame of the image handed to set_image
    "large": [dict(box=[100, 120, 500, 600], pts=None, labels=None), dict(box=[300, 80, 700, 400], pts=[[350, 200], [600, 300]], labels=[1, 1]),
              dict(box=None, pts=[[378, 378]], labels=[1])],
    "tiny": [dict(box=[30, 40, 120, 150], pts=None, labels=None), dict(box=[20, 20, 170, 100], pts=[[60, 70], [100, 90]], labels=[1, 0]),
             dict(box=None, pts=[[90, 95]], labels=[1])],
}


def hf_model(g: S.Sam2Geometry, W: dict, dtype=torch.float32):
    from transformers.models.sam2.configuration_sam2 import (Sam2Config, Sam2HieraDetConfig, Sam2MaskDecoderConfig, Sam2PromptEncoderConfig,
                                                              Sam2VisionConfig)
    from transformers.models.sam2.modeling_sam2 import Sam2Model
    bb = Sam2HieraDetConfig(hidden_size=g.embed_dims[0], num_attention_heads=g.heads[0], blocks_per_stage=list(g.blocks),
                            embed_dim_per_stage=list(g.embed_dims), num_attention_heads_per_stage=list(g.heads),
                            window_size_per_stage=list(g.windows), global_attention_blocks=list(g.global_blocks),
                            image_size=[g.image_size, g.image_size], window_positional_embedding_background_size=[g.bkg_size, g.bkg_size])
    m = g.image_size // 16
    vc = Sam2VisionConfig(backbone_config=bb, backbone_channel_list=list(reversed(g.embed_dims)),
                          backbone_feature_sizes=[[4 * m, 4 * m], [2 * m, 2 * m], [m, m]], fpn_hidden_size=g.fpn_dim)
    c = Sam2Config(vision_config=vc, prompt_encoder_config=Sam2PromptEncoderConfig(hidden_size=g.fpn_dim, image_size=g.image_size),
                   mask_decoder_config=Sam2MaskDecoderConfig(hidden_size=g.fpn_dim, mlp_dim=g.dec_mlp, num_hidden_layers=g.dec_layers,
                                                             num_attention_heads=g.dec_heads, iou_head_hidden_dim=g.fpn_dim))
    c._attn_implementation = "eager"
    model = Sam2Model(c).eval()
    sd = model.state_dict()
    missing = [k for k in sd if k not in W and "mask_embed" not in k]
    assert not missing, missing[:5]
    for k in sd:
        if k in W:
            sd[k].copy_(W[k].reshape(sd[k].shape))
    return model.to(dtype)


def bits(t):
    return t.detach().to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def run(tag, g, out):
    W = S.synthetic_weights(g)
    hw = 756 if tag == "large" else 189
    img = synthetic.tile_pixels(7, hw, hw)
    out[f"{tag}_img_seed"], out[f"{tag}_hw"] = np.array([7]), np.array([hw])
    st = STRIDE[tag]
    for dt, sfx in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        t0 = time.time()
        model = hf_model(g, W, dt)
        px = S.preprocess(img, g.image_size, dt)
        with torch.no_grad():
            vo = model.vision_encoder.backbone(px)
            for i, h in enumerate(vo.intermediate_hidden_states):
                v = h[0].flatten()[::st]
                out[f"{tag}_stage{i}_{sfx}"] = v.float().numpy() if dt == torch.float32 else bits(v)
            emb = model.get_image_embeddings(px)
            for i, f in enumerate(emb):
                v = f[0].permute(1, 2, 0).flatten()[::st]
                out[f"{tag}_feat{i}_{sfx}"] = v.float().numpy() if dt == torch.float32 else bits(v)
            for p, pr in enumerate(PROMPTS[tag]):
                c, l = S.prompt_points(pr["box"], pr["pts"], pr["labels"], (hw, hw), g.image_size)
                o = model(image_embeddings=emb, input_points=c[None, None].to(dt), input_labels=l[None, None].int(), multimask_output=True)
                low, iou = o.pred_masks[0, 0], o.iou_scores[0, 0]
                if dt == torch.float32:
                    out[f"{tag}_p{p}_box"] = np.array(pr["box"] if pr["box"] is not None else [], dtype=np.float32)
                    out[f"{tag}_p{p}_pts"] = np.array(pr["pts"] if pr["pts"] is not None else [], dtype=np.float32).reshape(-1, 2)
                    out[f"{tag}_p{p}_labels"] = np.array(pr["labels"] if pr["labels"] is not None else [], dtype=np.int64)
                    out[f"{tag}_p{p}_low"] = low.numpy().astype(np.float32)
                    out[f"{tag}_p{p}_iou"] = iou.numpy().astype(np.float32)
                    best, masks, up = S.postprocess(low, iou, (hw, hw))
                    out[f"{tag}_p{p}_best"] = np.array([int(torch.argmax(iou))])
                    out[f"{tag}_p{p}_mask_bits"] = np.packbits(best.numpy())
                    print(f"{tag} p{p}: |logit|max {float(low.abs().max()):.2f} rms {float(low.pow(2).mean().sqrt()):.2f} iou {iou.numpy()} mask area {int(best.sum())}")
                else:
                    out[f"{tag}_p{p}_low_bf16"] = bits(low)
                    out[f"{tag}_p{p}_iou_bf16"] = bits(iou)
                    ref = torch.from_numpy(out[f"{tag}_p{p}_low"])
                    d = low.float() - ref
                    print(f"{tag} p{p}: HF-bf16 vs f32 low-res logits max {float(d.abs().max()):.3f} rms {float(d.pow(2).mean().sqrt()):.4f}; "
                          f"iou {iou.float().numpy()} vs {out[f'{tag}_p{p}_iou']}; sign flips {int(((low.float() > 0) != (ref > 0)).sum())} of {ref.numel()}")
        print(f"{tag} {sfx}: {time.time() - t0:.1f}s", flush=True)


def main():
    torch.manual_seed(0)
    out = {"stride_tiny": np.array([STRIDE["tiny"]]), "stride_large": np.array([STRIDE["large"]])}
    run("tiny", S.geometry_tiny(), out)
    run("large", S.geometry_large(), out)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()
