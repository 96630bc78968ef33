"""CPU restatement of the SAM2 (Hiera) image path the reference's ``seg_infer`` role runs.  TEST INFRASTRUCTURE ONLY (imported by
tests/, tools/make_golden_sam2.py and nothing in the product).

Reference call sites: ``SegInferStrategy.segment`` (/root/reference/roll/distributed/strategy/seg_strategy.py:26-72) resizes the sample
to 756 x 756, calls ``SAM2ImagePredictor.set_image`` once and ``predict(point_coords, point_labels, box)`` per object (default
``multimask_output=True``), keeps ``pred_masks[argmax(scores)]`` and ORs the objects; the model is ``facebook/sam2-hiera-large`` built by
``sam2_seg_model_provider`` (/root/reference/roll/models/model_providers.py:515-562) and runs in float32 (no autocast anywhere on that
path).  The ``sam2`` package is NOT installed here and is not vendored by the reference; the arithmetic restated below is the one of
``transformers.models.sam2.modeling_sam2`` (transformers 5.15.0 -- the same network, HF parameter names), which tools/make_golden_sam2.py
executes to produce the fixtures this file is pinned against (tests/test_oracle_golden.py).  Line numbers ``hf:`` refer to that file.
What is restated from the sam2 package's published behaviour without being executable here ("unpinned"): the
predictor's pre/post-processing -- resize to 1024 x 1024 (bilinear, an upscale: antialiasing does not apply) and ImageNet normalisation,
prompt coordinates scaled by 1024 / 756, a box passed as two corner points labelled 2 / 3 IN FRONT of the click points plus the padding
point the prompt encoder appends, mask logits resized to 756 x 756 (bilinear, align_corners = False) and thresholded at 0.

All functions take the weights as a dict of HF-named tensors and compute in the dtype of those tensors (float32 = the reference's
numerics; bfloat16 = what a bf16 device path may be held to)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@dataclass
class Sam2Geometry:
    """Hiera-L + SAM2 decoder (sam2_hiera_l.yaml of the sam2 package: embed 144, heads 2, stages (2, 6, 36, 4), global blocks
    23 / 33 / 43, window spec (8, 4, 16, 8), 7 x 7 background position grid, FPN 256 with top-down levels 2 and 3)."""
    image_size: int = 1024
    embed_dims: Tuple[int, ...] = (144, 288, 576, 1152)
    heads: Tuple[int, ...] = (2, 4, 8, 16)
    blocks: Tuple[int, ...] = (2, 6, 36, 4)
    windows: Tuple[int, ...] = (8, 4, 16, 8)
    global_blocks: Tuple[int, ...] = (23, 33, 43)
    bkg_size: int = 7
    fpn_dim: int = 256
    top_down_levels: Tuple[int, ...] = (2, 3)
    dec_heads: int = 8
    dec_mlp: int = 2048
    dec_layers: int = 2
    n_mask_tokens: int = 4
    ln_eps: float = 1e-6

    def block_table(self):
        """[(stage, dim_in, dim_out, heads, window (0 = global), pooled)] for every block (hf:457-501)."""
        out, k = [], 0
        for s, n in enumerate(self.blocks):
            for b in range(n):
                first = s > 0 and b == 0
                win = self.windows[s - 1] if first else self.windows[s]
                if k in self.global_blocks:
                    win = 0
                out.append((s, self.embed_dims[s - 1] if first else self.embed_dims[s], self.embed_dims[s], self.heads[s], win, first))
                k += 1
        return out


def geometry_large() -> Sam2Geometry:
    return Sam2Geometry()


def geometry_tiny() -> Sam2Geometry:
    """Same structure, small: every kind of block occurs (windowed, pooled stage entry, global), windows divide the grids."""
    return Sam2Geometry(image_size=256, embed_dims=(32, 64, 128, 256), heads=(1, 2, 4, 8), blocks=(1, 2, 3, 2), windows=(8, 4, 8, 4),
                        global_blocks=(4,), bkg_size=7, fpn_dim=256, dec_mlp=256)


# ------------------------------------------------------------------------------------------------ synthetic weights
def param_specs(g: Sam2Geometry) -> List[Tuple[str, tuple, float, float]]:
    """(HF name, shape, base, std).  Real weights are not available offline: linear / conv weights ~ N(0, 1 / fan_in) keep the
    activations of order one through all 48 blocks, so that mask logits are not numerically trivial."""
    sp: List[Tuple[str, tuple, float, float]] = []

    def lin(name, out_f, in_f, k=1):
        sp.append((name + ".weight", (out_f, in_f) if k == 0 else (out_f, in_f, k, k), 0.0, 1.0 / math.sqrt(in_f * max(k, 1) ** 2)))
        sp.append((name + ".bias", (out_f,), 0.0, 0.02))

    def ln(name, c):
        sp.append((name + ".weight", (c,), 1.0, 0.02))
        sp.append((name + ".bias", (c,), 0.0, 0.02))

    d0 = g.embed_dims[0]
    sp.append(("no_memory_embedding", (1, 1, g.fpn_dim), 0.0, 0.02))
    sp.append(("shared_image_embedding.positional_embedding", (2, g.fpn_dim // 2), 0.0, 1.0))
    sp.append(("vision_encoder.backbone.pos_embed", (1, d0, g.bkg_size, g.bkg_size), 0.0, 0.02))
    sp.append(("vision_encoder.backbone.pos_embed_window", (1, d0, g.windows[0], g.windows[0]), 0.0, 0.02))
    lin("vision_encoder.backbone.patch_embed.projection", d0, 3, 7)
    for i, (s, din, dout, heads, win, pooled) in enumerate(g.block_table()):
        p = f"vision_encoder.backbone.blocks.{i}"
        ln(p + ".layer_norm1", din)
        lin(p + ".attn.qkv", 3 * dout, din, 0)
        lin(p + ".attn.proj", dout, dout, 0)
        ln(p + ".layer_norm2", dout)
        lin(p + ".mlp.proj_in", 4 * dout, dout, 0)
        lin(p + ".mlp.proj_out", dout, 4 * dout, 0)
        if din != dout:
            lin(p + ".proj", dout, din, 0)
    for j, c in enumerate(reversed(g.embed_dims)):
        lin(f"vision_encoder.neck.convs.{j}", g.fpn_dim, c, 1)
    C = g.fpn_dim
    sp.append(("prompt_encoder.shared_embedding.positional_embedding", (2, C // 2), 0.0, 1.0))     # (tied to shared_image_embedding)
    sp.append(("prompt_encoder.no_mask_embed.weight", (1, C), 0.0, 0.5))
    sp.append(("prompt_encoder.point_embed.weight", (4, C), 0.0, 0.5))
    sp.append(("prompt_encoder.not_a_point_embed.weight", (1, C), 0.0, 0.5))
    sp.append(("mask_decoder.iou_token.weight", (1, C), 0.0, 0.5))
    sp.append(("mask_decoder.mask_tokens.weight", (g.n_mask_tokens, C), 0.0, 0.5))
    sp.append(("mask_decoder.obj_score_token.weight", (1, C), 0.0, 0.5))

    def attn(name, internal):
        for q in ("q_proj", "k_proj", "v_proj"):
            lin(f"{name}.{q}", internal, C, 0)
        lin(f"{name}.o_proj", C, internal, 0)

    for l in range(g.dec_layers):
        p = f"mask_decoder.transformer.layers.{l}"
        attn(p + ".self_attn", C)
        ln(p + ".layer_norm1", C)
        attn(p + ".cross_attn_token_to_image", C // 2)
        ln(p + ".layer_norm2", C)
        lin(p + ".mlp.proj_in", g.dec_mlp, C, 0)
        lin(p + ".mlp.proj_out", C, g.dec_mlp, 0)
        ln(p + ".layer_norm3", C)
        ln(p + ".layer_norm4", C)
        attn(p + ".cross_attn_image_to_token", C // 2)
    attn("mask_decoder.transformer.final_attn_token_to_image", C // 2)
    ln("mask_decoder.transformer.layer_norm_final_attn", C)
    sp.append(("mask_decoder.upscale_conv1.weight", (C, C // 4, 2, 2), 0.0, 1.0 / math.sqrt(C)))
    sp.append(("mask_decoder.upscale_conv1.bias", (C // 4,), 0.0, 0.02))
    sp.append(("mask_decoder.upscale_conv2.weight", (C // 4, C // 8, 2, 2), 0.0, 1.0 / math.sqrt(C // 4)))
    sp.append(("mask_decoder.upscale_conv2.bias", (C // 8,), 0.0, 0.02))
    ln("mask_decoder.upscale_layer_norm", C // 4)

    def mlp3(name, hid, out_f):
        lin(name + ".proj_in", hid, C, 0)
        lin(name + ".layers.0", hid, hid, 0)
        lin(name + ".proj_out", out_f, hid, 0)

    for i in range(g.n_mask_tokens):
        mlp3(f"mask_decoder.output_hypernetworks_mlps.{i}", C, C // 8)
    mlp3("mask_decoder.iou_prediction_head", C, g.n_mask_tokens)
    lin("mask_decoder.conv_s0", C // 8, C, 1)
    lin("mask_decoder.conv_s1", C // 4, C, 1)
    mlp3("mask_decoder.pred_obj_score_head", C, 1)
    return sp


def synthetic_weights(g: Sam2Geometry, seed: int = 0, dtype=torch.float32) -> dict:
    """bf16-representable values from the repo's counter-based generator (oracle/weights.py), scaled per tensor."""
    from oracle import weights as WG
    out = {}
    for name, shape, base, std in param_specs(g):
        src = "shared_image_embedding.positional_embedding" if name.endswith("shared_embedding.positional_embedding") else name
        z = WG.synth_f32_torch(src, shape, seed=seed, base=0.0) * (1.0 / 0.02)          # ~ N(0, 1), a multiple of the generator's step
        out[name] = (base + z * std).to(torch.bfloat16).to(dtype).reshape(shape)
    return out


# ------------------------------------------------------------------------------------------------ predictor pre / post-processing
def preprocess(img_u8: np.ndarray, size: int = 1024, dtype=torch.float32) -> torch.Tensor:
    """uint8 HWC -> float [1, 3, size, size]: /255, bilinear resize (align_corners False), ImageNet normalisation."""
    x = torch.from_numpy(np.array(img_u8, dtype=np.uint8, order="C")).permute(2, 0, 1)[None].float() / 255.0
    x = F.interpolate(x, size=(size, size), mode="bilinear", align_corners=False)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    return ((x - mean) / std).to(dtype)


def prompt_points(box: Optional[Sequence[float]], points, labels, orig_hw: Tuple[int, int], size: int = 1024):
    """The predictor's prompt: coordinates scaled to the model's input frame; a box becomes two corner points labelled 2 and 3 placed in
    front of the click points.  -> (coords float32 [P, 2], labels int64 [P])."""
    h, w = orig_hw
    cs, ls = [], []
    if box is not None:
        b = np.asarray(box, dtype=np.float32).reshape(2, 2)
        cs.append(b)
        ls.append(np.array([2, 3], dtype=np.int64))
    if points is not None and len(points):
        cs.append(np.asarray(points, dtype=np.float32).reshape(-1, 2))
        ls.append(np.asarray(labels, dtype=np.int64).reshape(-1))
    c = np.concatenate(cs, axis=0).astype(np.float32)
    c = c * np.array([size / w, size / h], dtype=np.float32)
    return torch.from_numpy(c), torch.from_numpy(np.concatenate(ls))


def postprocess(low_res: torch.Tensor, iou: torch.Tensor, orig_hw: Tuple[int, int]):
    """low_res [3, m, m] mask logits, iou [3] -> (best mask uint8 [h, w], masks uint8 [3, h, w], logits float32 [3, h, w])."""
    up = F.interpolate(low_res[None].float(), size=orig_hw, mode="bilinear", align_corners=False)[0]
    masks = (up > 0.0).to(torch.uint8)
    return masks[int(torch.argmax(iou.float()))], masks, up


# ------------------------------------------------------------------------------------------------ image encoder (Hiera + FPN neck)
def _ln(x, W, name, eps):
    return F.layer_norm(x, (x.shape[-1],), W[name + ".weight"], W[name + ".bias"], eps)


def _lin(x, W, name):
    return F.linear(x, W[name + ".weight"], W[name + ".bias"])


def _windows(x, ws):
    """[H, W, C] -> [nH * nW, ws * ws, C]  (hf:397-425; sizes divide)."""
    H, Wd, C = x.shape
    return x.view(H // ws, ws, Wd // ws, ws, C).permute(0, 2, 1, 3, 4).reshape(-1, ws * ws, C)


def _unwindows(w, ws, H, Wd):
    C = w.shape[-1]
    return w.view(H // ws, Wd // ws, ws, ws, C).permute(0, 2, 1, 3, 4).reshape(H, Wd, C)


def _pool(x):
    """2 x 2 max pooling of [H, W, C] (hf:290-298)."""
    return F.max_pool2d(x.permute(2, 0, 1)[None].float(), 2, 2)[0].permute(1, 2, 0).to(x.dtype)


def _attend(q, k, v, scale):
    """[..., heads, nq, d] x [..., heads, nk, d]: scores * scale, float32 softmax, cast back, times V (hf:268-287)."""
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(p, v)


def hiera_pos_embed(W, g: Sam2Geometry, hw: int) -> torch.Tensor:
    """[hw, hw, C]: bicubic-resized background grid + tiled window embedding (hf:645-651)."""
    pe = W["vision_encoder.backbone.pos_embed"]
    win = W["vision_encoder.backbone.pos_embed_window"]
    p = F.interpolate(pe.float(), size=(hw, hw), mode="bicubic").to(pe.dtype)
    p = p + win.tile(1, 1, hw // win.shape[2], hw // win.shape[3])
    return p[0].permute(1, 2, 0)


def hiera_forward(W, g: Sam2Geometry, pixels: torch.Tensor, capture=None) -> List[torch.Tensor]:
    """pixels [1, 3, S, S] -> the 4 stage outputs [H_s, W_s, C_s] (hf:653-675, 490-552, 324-364)."""
    x = F.conv2d(pixels, W["vision_encoder.backbone.patch_embed.projection.weight"], W["vision_encoder.backbone.patch_embed.projection.bias"],
                 stride=4, padding=3)[0].permute(1, 2, 0)
    x = x + hiera_pos_embed(W, g, x.shape[0])
    ends = set(np.cumsum(g.blocks) - 1)
    outs = []
    for i, (s, din, dout, heads, win, pooled) in enumerate(g.block_table()):
        p = f"vision_encoder.backbone.blocks.{i}"
        H, Wd, _ = x.shape
        h = _ln(x, W, p + ".layer_norm1", g.ln_eps)
        res = _pool(_lin(h, W, p + ".proj")) if din != dout else x
        tok = _windows(h, win) if win > 0 else h.reshape(1, H * Wd, din)
        qkv = _lin(tok, W, p + ".attn.qkv").reshape(tok.shape[0], tok.shape[1], 3, heads, dout // heads)
        q, k, v = qkv.unbind(2)                                         # [nw, n, heads, d]
        if pooled:                                                      # queries are max-pooled 2 x 2 inside their window
            ws = win if win > 0 else H
            q = q.reshape(q.shape[0], ws, ws, dout)
            q = torch.stack([_pool(qq) for qq in q]).reshape(q.shape[0], (ws // 2) ** 2, heads, dout // heads)
        scale = (dout // heads) ** -0.5
        o = _attend(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), scale).permute(0, 2, 1, 3).reshape(q.shape[0], q.shape[1], dout)
        o = _lin(o, W, p + ".attn.proj")
        Ho, Wo = (H // 2, Wd // 2) if pooled else (H, Wd)
        o = _unwindows(o, (win // 2 if pooled else win), Ho, Wo) if win > 0 else o.reshape(Ho, Wo, dout)
        x = res + o
        x = x + _lin(F.gelu(_lin(_ln(x, W, p + ".layer_norm2", g.ln_eps), W, p + ".mlp.proj_in")), W, p + ".mlp.proj_out")
        if capture is not None:
            capture(i, x)
        if i in ends:
            outs.append(x)
    return outs


def neck_forward(W, g: Sam2Geometry, stages: List[torch.Tensor]) -> List[torch.Tensor]:
    """FPN (hf:216-265) + the decoder's conv_s0 / conv_s1 and the no-memory embedding (hf:1586-1611, 1499-1506).
    -> [feat_s0 [4m, 4m, C/8], feat_s1 [2m, 2m, C/4], image_embed [m, m, C]]  (m = image_size / 16)."""
    n = len(stages) - 1
    prev, fpn = None, {}
    for i in range(n, -1, -1):
        lat = _lin(stages[i], W, f"vision_encoder.neck.convs.{n - i}")
        if i in g.top_down_levels and i != n:
            up = F.interpolate(prev.permute(2, 0, 1)[None].float(), scale_factor=2.0, mode="nearest")[0].permute(1, 2, 0).to(lat.dtype)
            prev = lat + up
        else:
            prev = lat
        fpn[i] = prev
    f0 = _lin(fpn[0], W, "mask_decoder.conv_s0")
    f1 = _lin(fpn[1], W, "mask_decoder.conv_s1")
    emb = fpn[2] + W["no_memory_embedding"].reshape(-1)
    return [f0, f1, emb]


def _conv1x1_weights(W):
    """1 x 1 conv weights act as Linear weights on channel-last tensors."""
    for k in list(W):
        if W[k].dim() == 4 and W[k].shape[2:] == (1, 1) and ("neck.convs" in k or "conv_s" in k):
            W[k] = W[k].reshape(W[k].shape[0], W[k].shape[1])
    return W


# ------------------------------------------------------------------------------------------------ prompt encoder + mask decoder
def _pe(W, coords01: torch.Tensor) -> torch.Tensor:
    """random-Fourier position encoding of coordinates in [0, 1] (hf:727-749)."""
    G = W["prompt_encoder.shared_embedding.positional_embedding"]
    c = (2 * coords01 - 1).to(G.dtype) @ G
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def prompt_encode(W, g: Sam2Geometry, coords: torch.Tensor, labels: torch.Tensor):
    """coords [P, 2] in input-frame pixels, labels [P] (1 / 0 click, 2 / 3 box corners) -> sparse [P + 1, C]: the padding point the encoder
    appends when no box tensor is passed (hf:791-813), point / corner label embeddings added."""
    dt = W["prompt_encoder.point_embed.weight"].dtype
    pts = torch.cat([coords.to(dt) + 0.5, torch.zeros(1, 2, dtype=dt)], dim=0)
    lab = torch.cat([labels, torch.tensor([-1])])
    e = _pe(W, pts / g.image_size)
    e = torch.where(lab[:, None] == -1, W["prompt_encoder.not_a_point_embed.weight"], e)
    e = e + W["prompt_encoder.point_embed.weight"][lab.clamp(min=0)] * (lab >= 0)[:, None].to(dt)
    return e


def image_pe(W, g: Sam2Geometry) -> torch.Tensor:
    """[m * m, C] dense position encoding of the embedding grid (hf:1354-1364)."""
    m = g.image_size // 16
    dt = W["prompt_encoder.shared_embedding.positional_embedding"].dtype
    ax = (torch.arange(m, dtype=dt) + 0.5) / m
    yy, xx = torch.meshgrid(ax, ax, indexing="ij")
    return _pe(W, torch.stack([xx, yy], dim=-1)).reshape(m * m, -1)


def _mha(W, name, q, k, v, heads):
    """Sam2Attention (hf:874-942): projections to the internal width, per-head attention, output projection."""
    qp, kp, vp = _lin(q, W, name + ".q_proj"), _lin(k, W, name + ".k_proj"), _lin(v, W, name + ".v_proj")
    d = qp.shape[-1] // heads
    sp = lambda t: t.reshape(t.shape[0], heads, d).transpose(0, 1)
    o = _attend(sp(qp), sp(kp), sp(vp), d ** -0.5).transpose(0, 1).reshape(q.shape[0], heads * d)
    return _lin(o, W, name + ".o_proj")


def _mlp(W, name, x, n_hidden_layers=1, act=F.relu, sigmoid=False):
    x = act(_lin(x, W, name + ".proj_in"))
    for j in range(n_hidden_layers):
        x = act(_lin(x, W, f"{name}.layers.{j}"))
    x = _lin(x, W, name + ".proj_out")
    return torch.sigmoid(x) if sigmoid else x


def mask_decode(W, g: Sam2Geometry, feats: List[torch.Tensor], sparse: torch.Tensor, capture=None):
    """-> (low-res mask logits [3, 4m, 4m], iou [3], object score logit [1])  with multimask_output = True (hf:1145-1254, 945-1076)."""
    f0, f1, emb = feats
    m, C = emb.shape[0], emb.shape[-1]
    eps = 1e-5                                                           # nn.LayerNorm default inside the decoder (hf:965-976, 1033)
    tokens = torch.cat([W["mask_decoder.obj_score_token.weight"], W["mask_decoder.iou_token.weight"], W["mask_decoder.mask_tokens.weight"], sparse], dim=0)
    keys = (emb + W["prompt_encoder.no_mask_embed.weight"].reshape(-1)).reshape(m * m, C)
    kpe, qpe = image_pe(W, g), tokens
    queries = tokens
    for l in range(g.dec_layers):
        p = f"mask_decoder.transformer.layers.{l}"
        if l == 0:
            queries = _mha(W, p + ".self_attn", queries, queries, queries, g.dec_heads)
        else:
            q = queries + qpe
            queries = queries + _mha(W, p + ".self_attn", q, q, queries, g.dec_heads)
        queries = _ln(queries, W, p + ".layer_norm1", eps)
        queries = queries + _mha(W, p + ".cross_attn_token_to_image", queries + qpe, keys + kpe, keys, g.dec_heads)
        queries = _ln(queries, W, p + ".layer_norm2", eps)
        queries = queries + _mlp(W, p + ".mlp", queries, 0)
        queries = _ln(queries, W, p + ".layer_norm3", eps)
        keys = keys + _mha(W, p + ".cross_attn_image_to_token", keys + kpe, queries + qpe, queries, g.dec_heads)
        keys = _ln(keys, W, p + ".layer_norm4", eps)
        if capture is not None:
            capture(f"dec{l}", queries, keys)
    queries = queries + _mha(W, "mask_decoder.transformer.final_attn_token_to_image", queries + qpe, keys + kpe, keys, g.dec_heads)
    queries = _ln(queries, W, "mask_decoder.transformer.layer_norm_final_attn", eps)
    # upscaling: two stride-2 transposed convolutions with the high-resolution features added (hf:1215-1221)
    src = keys.reshape(m, m, C).permute(2, 0, 1)[None]
    up = F.conv_transpose2d(src, W["mask_decoder.upscale_conv1.weight"], W["mask_decoder.upscale_conv1.bias"], stride=2) + f1.permute(2, 0, 1)[None]
    up = F.layer_norm(up.permute(0, 2, 3, 1), (C // 4,), W["mask_decoder.upscale_layer_norm.weight"], W["mask_decoder.upscale_layer_norm.bias"], 1e-6)
    up = F.gelu(up).permute(0, 3, 1, 2)
    up = F.gelu(F.conv_transpose2d(up, W["mask_decoder.upscale_conv2.weight"], W["mask_decoder.upscale_conv2.bias"], stride=2) + f0.permute(2, 0, 1)[None])
    hyper = torch.stack([_mlp(W, f"mask_decoder.output_hypernetworks_mlps.{i}", queries[2 + i:3 + i], 1)[0] for i in range(g.n_mask_tokens)])
    masks = (hyper @ up[0].reshape(C // 8, -1)).reshape(g.n_mask_tokens, 4 * m, 4 * m)
    iou = _mlp(W, "mask_decoder.iou_prediction_head", queries[1:2], 1, sigmoid=True)[0]
    obj = _mlp(W, "mask_decoder.pred_obj_score_head", queries[0:1], 1)[0]
    return masks[1:], iou[1:], obj


# ------------------------------------------------------------------------------------------------ the predictor
class Sam2Oracle:
    """set_image / predict with the reference predictor's contract: predict(point_coords, point_labels, box) -> (masks [3, h, w] bool,
    scores [3], low-res logits [3, 4m, 4m])."""

    def __init__(self, W: dict, g: Sam2Geometry):
        self.W, self.g = _conv1x1_weights(dict(W)), g
        self.dtype = W["no_memory_embedding"].dtype

    def set_image(self, img_u8: np.ndarray):
        self.orig_hw = img_u8.shape[:2]
        with torch.no_grad():
            px = preprocess(img_u8, self.g.image_size, self.dtype)
            self.stages = hiera_forward(self.W, self.g, px)
            self.feats = neck_forward(self.W, self.g, self.stages)

    def predict(self, point_coords=None, point_labels=None, box=None):
        with torch.no_grad():
            c, l = prompt_points(box, point_coords, point_labels, self.orig_hw, self.g.image_size)
            sparse = prompt_encode(self.W, self.g, c, l)
            low, iou, obj = mask_decode(self.W, self.g, self.feats, sparse)
            best, masks, up = postprocess(low, iou, self.orig_hw)
        self.last = {"low_res": low.float(), "iou": iou.float(), "obj": obj.float(), "logits": up}
        return masks.bool().numpy(), iou.float().numpy(), low.float().numpy()
