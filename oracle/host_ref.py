"""Oracle for the integer / host-side rows of the hot path.
TEST INFRASTRUCTURE ONLY.  numpy / torch-CPU restatements; each function cites what it follows.
``hf:`` = site-packages/transformers (5.15.0), other paths are relative to /root/reference.
"""
from __future__ import annotations

import json
import math
import re

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ----------------------------------------------------------------------------- A3 / A4 image side
def smart_resize(height, width, factor=28, min_pixels=56 * 56, max_pixels=768 * 768):
    """hf:models/qwen2_vl/image_processing_pil_qwen2_vl.py:57-83; defaults as the reference binds them at
    roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:518-521 (the YAML max/min_pixels are dropped)."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError("absolute aspect ratio must be smaller than 200")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def normalize_u8(img_hwc: np.ndarray) -> np.ndarray:
    """hf:image_transforms.py:89-124 (float64 multiply, float32 downcast) and :384-440 (float32 (x-mean)/std)."""
    x = (img_hwc.astype(np.float64) * (1 / 255)).astype(np.float32)
    mean = np.array(CLIP_MEAN, dtype=np.float32)
    std = np.array(CLIP_STD, dtype=np.float32)
    return (x - mean) / std  # HWC, float32


def patchify(img_hwc_u8: np.ndarray, patch=14, merge=2, temporal=2):
    """hf:models/qwen2_vl/image_processing_pil_qwen2_vl.py:153-185: rows (gh/m, gw/m, m, m), cols (C, T, p, p);
    the single frame is duplicated along T.  Returns (float32 [N, C*T*p*p], (1, gh, gw))."""
    x = normalize_u8(img_hwc_u8).transpose(2, 0, 1)  # CHW
    c, h, w = x.shape
    gh, gw = h // patch, w // patch
    p = x.reshape(c, gh // merge, merge, patch, gw // merge, merge, patch)
    p = p.transpose(1, 4, 2, 5, 0, 3, 6)
    p = np.broadcast_to(p[:, :, :, :, :, None, :, :], (*p.shape[:5], temporal, *p.shape[5:]))
    return np.ascontiguousarray(p.reshape(gh * gw, c * temporal * patch * patch)), (1, gh, gw)


# ----------------------------------------------------------------------------- A5 mRoPE position ids
def get_rope_index(input_ids: np.ndarray, image_grid_thw, attention_mask=None, *, merge=2,
                   image_token_id=151655, vision_start_token_id=151652):
    """mcore_adapter/src/mcore_adapter/models/qwen2_5_vl/modeling_qwen2_5_vl.py:319-441 (image-only branch; the
    infer pipeline has no videos).  input_ids [B,S] int64 -> (position_ids [3,B,S] int64, deltas [B,1])."""
    input_ids = np.asarray(input_ids, dtype=np.int64)
    B, S = input_ids.shape
    if attention_mask is None:
        attention_mask = np.ones_like(input_ids)
    attention_mask = np.asarray(attention_mask)
    if image_grid_thw is None or len(image_grid_thw) == 0:
        pos = np.cumsum(attention_mask.astype(np.int64), -1) - 1
        pos[attention_mask == 0] = 1
        pos3 = np.broadcast_to(pos[None], (3, B, S)).copy()
        deltas = pos3.max(0).max(-1, keepdims=True) + 1 - S
        return pos3, deltas
    pos3 = np.ones((3, B, S), dtype=np.int64)
    deltas = []
    img_i = 0
    for b in range(B):
        ids = input_ids[b][attention_mask[b] == 1]
        toks = ids.tolist()
        starts = np.nonzero(ids == vision_start_token_id)[0]
        n_img = int((ids[starts + 1] == image_token_id).sum()) if len(starts) else 0
        chunks, st = [], 0
        for _ in range(n_img):
            ed = toks.index(image_token_id, st)
            t, h, w = (int(v) for v in image_grid_thw[img_i])
            img_i += 1
            lt, lh, lw = t, h // merge, w // merge
            text_len = ed - st
            st_idx = int(chunks[-1].max()) + 1 if chunks else 0
            chunks.append(np.broadcast_to(np.arange(text_len)[None], (3, text_len)) + st_idx)
            ti = np.repeat(np.arange(lt), lh * lw) * 0  # second_per_grid_t = 0 for images
            hi = np.tile(np.repeat(np.arange(lh), lw), lt)
            wi = np.tile(np.arange(lw), lt * lh)
            chunks.append(np.stack([ti, hi, wi]) + text_len + st_idx)
            st = ed + lt * lh * lw
        if st < len(toks):
            st_idx = int(chunks[-1].max()) + 1 if chunks else 0
            tl = len(toks) - st
            chunks.append(np.broadcast_to(np.arange(tl)[None], (3, tl)) + st_idx)
        llm = np.concatenate(chunks, axis=1).reshape(3, -1)
        pos3[:, b, attention_mask[b] == 1] = llm
        deltas.append(int(llm.max()) + 1 - S)
    return pos3, np.array(deltas, dtype=np.int64)[:, None]


# ----------------------------------------------------------------------------- A6 / A8 output layout
def gather_outputs_to_pad_tensor(token_lists, pad_token_id):
    """roll/distributed/strategy/vllm_strategy.py:279-286 (pad_sequence, batch_first, right pad)."""
    L = max((len(t) for t in token_lists), default=0)
    out = np.full((len(token_lists), L), pad_token_id, dtype=np.int64)
    for i, t in enumerate(token_lists):
        out[i, : len(t)] = t
    return out


def concatenate_input_and_output(input_ids, output_ids, num_return_sequences=1):
    """roll/utils/functionals.py:364-373."""
    rep = np.repeat(np.asarray(input_ids), num_return_sequences, axis=0)
    return np.concatenate([rep, np.asarray(output_ids)], axis=1)


def postprocess_generate(input_ids, attention_mask, position_ids, output, num_return_sequences, sequence_length,
                         eos_token_id, pad_token_id, fill_eos_token=False):
    """roll/utils/functionals.py:768-872, mRoPE (3-D position_ids) branch.  numpy in, dict of numpy out."""
    output = np.array(output, dtype=np.int64, copy=True)
    if fill_eos_token:
        last = output.shape[1] - 1
        need = output[:, last] != pad_token_id
        output[need, last] = eos_token_id
    input_ids = np.asarray(input_ids)
    attention_mask = np.asarray(attention_mask)
    obs = output.shape[0]
    P = input_ids.shape[1]
    if output.shape[1] >= sequence_length:  # pad_to_length :351-361
        output = output[:, :sequence_length]
    else:
        pad = np.full((obs, sequence_length - output.shape[1]), pad_token_id, dtype=output.dtype)
        output = np.concatenate([output, pad], axis=1)
    prompt = output[:, :P].copy()
    response = output[:, P:].copy()
    attention_mask = np.repeat(attention_mask, num_return_sequences, axis=0)
    response_mask = (response != pad_token_id).astype(attention_mask.dtype)  # get_pad_mask :301-313
    assert not ((response_mask[:, 0] == 0) & (response_mask.sum(-1) != 0)).any()
    attention_mask = np.concatenate([attention_mask, response_mask], axis=-1)
    position_ids = np.repeat(np.asarray(position_ids), num_return_sequences, axis=0)  # [B,3,P]
    delta = np.arange(1, sequence_length - P + 1)[None, None, :]
    out_pos = np.concatenate([position_ids, position_ids[..., -1:] + delta], axis=-1)
    assert attention_mask.any(axis=1).all()
    first_one = attention_mask.astype(np.float32).argmax(axis=1)
    new_response_mask = np.zeros_like(attention_mask)
    for i in range(obs):
        shift = int(first_one[i])
        if shift > 0:
            output[i, :-shift] = output[i, shift:].copy()
        valid = int(attention_mask[i].sum())
        rl = int(response_mask[i].sum())
        attention_mask[i][:valid] = 1
        attention_mask[i][valid:] = 0
        new_response_mask[i][valid - rl: valid] = 1
        if shift > 0:
            out_pos[i, ..., :-shift] = out_pos[i, ..., shift:].copy()
            if P > rl:
                output[i, -shift:] = pad_token_id
    prompt_mask = (attention_mask == 1) & (new_response_mask == 0)
    return {
        "prompts": prompt, "responses": response, "input_ids": output, "attention_mask": attention_mask,
        "position_ids": out_pos, "prompt_mask": prompt_mask, "response_mask": new_response_mask,
    }


# ----------------------------------------------------------------------------- A10 parsers
_ANSWER = re.compile(r"<answer>(.*?)</answer>", re.DOTALL)


def parse_points_text_from_content(content: str) -> str:
    """roll/pipeline/multi_utils.py:4-15."""
    m = _ANSWER.search(content)
    return m.group(1).strip() if m else ""


def parse_visual_prompt_from_json_s2(content: str):
    """roll/pipeline/rlvr/seg_worker.py:199-259."""
    out = []
    m = _ANSWER.search(content)
    if not m:
        return out
    try:
        data = json.loads(m.group(1).strip())
    except json.JSONDecodeError:
        return out
    if not isinstance(data, list):
        return []
    for obj in data:
        try:
            if not isinstance(obj, dict):
                continue
            box = obj.get("bbox_2d", [])
            pts = [[p[0], p[1]] for p in obj.get("points", [])]
            labels = [1] * len(pts)
            if isinstance(box, list) and len(box) == 4:
                out.append({"box": box, "points": pts, "labels": labels})
        except Exception:
            continue
    return out


# ----------------------------------------------------------------------------- A11-A13 raster (numpy mirror of raster_ref.c)
def mask_union(masks):
    """roll/distributed/strategy/seg_strategy.py:58-60: acc = logical_or(acc, m).astype(uint8)."""
    acc = np.zeros_like(masks[0], dtype=np.uint8)
    for m in masks:
        acc = np.logical_or(acc, m).astype(np.uint8)
    return acc


def resize_nearest(src: np.ndarray, H: int, W: int) -> np.ndarray:
    """cv2.INTER_NEAREST as called at seg_strategy.py:65 and rlvr_socioseg_vlm_pipeline_infer.py:399.
    cv2 is not installed here (parity of this rule is UNPINNED): documented rule
    sx = min(floor(dx * ifx), sw - 1) with ifx = 1 / (dw / sw) formed like cv::resize forms it (inv_scale = dsize / ssize, then
    its reciprocal) -- NOT floor(dx * sw / dw): the two differ for rare size pairs (e.g. 768 -> 1148)."""
    h, w = src.shape[:2]
    ys = np.minimum(np.floor(np.arange(H) * (1.0 / (H / h))).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(W) * (1.0 / (W / w))).astype(np.int64), w - 1)
    return src[ys][:, xs]


def iou_counts(pred: np.ndarray, gt: np.ndarray):
    """rlvr_socioseg_vlm_pipeline_infer.py:45-58 integer part."""
    p, g = pred > 0, gt > 0
    return int(np.logical_and(p, g).sum()), int(np.logical_or(p, g).sum())


def compute_giou(pred, gt) -> float:
    i, u = iou_counts(pred, gt)
    return 1.0 if u == 0 else i / u


INT_STANDIN = -(1 << 30)     # stands in for the INT_MIN an x86 double->int cast yields for NaN / out-of-range values


def pil_box(bb):
    """What PIL's ImageDraw.rectangle([(b0, b1), (b2, b3)]) does with one model-generated box before it draws
    (PIL 12.2 _imaging.c _draw_rectangle + path.c PyPath_Flatten; probed by tools/make_golden.py):
      * every coordinate must be a Python int / float (bool is an int); anything else -> ValueError -> the reference
        skips the box (rlvr_socioseg_vlm_pipeline_infer.py:427-433);
      * x1 < x0 or y1 < y0 -- compared as DOUBLES, before any truncation -> ValueError -> skipped;
      * then each coordinate is cast (int): truncation toward zero.
    Returns 4 ints or None (skipped).  NaN / |v| >= 2^31 become INT_MIN in the x86 cast; -2^30 draws the same pixels."""
    if not isinstance(bb, (list, tuple)) or len(bb) != 4:
        return None
    f = []
    for v in bb:
        if not isinstance(v, (bool, int, float)):
            return None
        try:
            f.append(float(v))
        except OverflowError:
            return None
    if f[2] < f[0] or f[3] < f[1]:
        return None
    return [int(v) if (v == v and abs(v) < 2147483648.0) else INT_STANDIN for v in f]


def render_overlay(img_rgb: np.ndarray, mask: np.ndarray, bboxes, alpha=102, color=(255, 0, 0)):
    """rlvr_socioseg_vlm_pipeline_infer.py:383-452 for one image, integer-exact restatement of the PIL calls:
    RGB->RGBA(A=255), ImageDraw.rectangle(outline=blue, width=2) per bbox, then
    Image.alpha_composite with an overlay that is (255,0,0,102) where mask>0 else (0,0,0,0), -> RGB.
    The rectangles are drawn BEFORE the overlay is composited (reference order)."""
    h, w = img_rgb.shape[:2]
    out = img_rgb.astype(np.int64).copy()
    for bb in bboxes:
        ib = pil_box(bb)
        if ib is None:
            continue  # PIL raises -> the reference swallows it (:431-432)
        draw_rect_outline(out, ib[0], ib[1], ib[2], ib[3], 2, (0, 0, 255))
    if mask is None:
        return out.astype(np.uint8)
    m = resize_nearest((mask > 0).astype(np.uint8), h, w) > 0
    # PIL AlphaComposite.c with dst.a == 255 (verified against PIL 12.2 in tools/make_golden.py):
    #   coef1 = a*128, coef2 = (255-a)*128, out = SHIFTFORDIV255(src*coef1 + dst*coef2 + (0x80<<7)) >> 7
    for c in range(3):
        t = (np.int64(color[c]) * alpha + out[..., c] * (255 - alpha)) * 128 + (128 << 7)
        blended = (((t >> 8) + t) >> 8) >> 7
        out[..., c] = np.where(m, blended, out[..., c])
    return out.astype(np.uint8)


def draw_rect_outline(img, x0, y0, x1, y1, width, color):
    """PIL ImageDraw.rectangle(outline, width) on integer coordinates (libImaging Draw.c, outline branch):
    per i < width: hline(x0, y0+i, x1), hline(x0, y1-i, x1), vertical lines at x1-i and x0+i from y0+width towards
    y1-width+1 (|dy| points from the start, end point excluded) -- so boxes thinner than 3 px spill outside themselves
    exactly as PIL's do.  Clipped to the image."""
    h, w = img.shape[:2]

    def span(a, b, n):
        a, b = (a, b) if a <= b else (b, a)
        return max(a, 0), min(b, n - 1)

    def hline(xa, y, xb):
        if 0 <= y < h:
            lo, hi = span(xa, xb, w)
            if hi >= lo:
                img[y, lo: hi + 1] = color

    def vline(x, ya, yb):
        # libImaging's vertical line loop draws |yb - ya| points starting at ya and stepping toward yb: the end point
        # itself is not drawn
        if ya == yb or not 0 <= x < w:
            return
        lo, hi = (ya, yb - 1) if yb > ya else (yb + 1, ya)
        lo, hi = max(lo, 0), min(hi, h - 1)
        if hi >= lo:
            img[lo: hi + 1, x] = color

    if y0 > y1:
        y0, y1 = y1, y0
    for i in range(width):
        hline(x0, y0 + i, x1)
        hline(x0, y1 - i, x1)
        vline(x1 - i, y0 + width, y1 - width + 1)
        vline(x0 + i, y0 + width, y1 - width + 1)


def render_image(bboxes_json: str, images, mask):
    """rlvr_socioseg_vlm_pipeline_infer.py:383-452 as a whole, on uint8 HWC arrays: the bbox list parsed the way the
    reference parses it (a non-list JSON value -> no boxes; items must be dicts holding a 4-element ``bbox_2d``; a
    ``bbox_2d`` without len() raises TypeError inside the same try -> ALL boxes dropped), ONE overlay at the first image's
    size (mask nearest-resized there), outlines drawn before compositing; an image of another size gets the overlay
    resampled with PIL's LANCZOS (:436-438) -- that resample is PIL's own (this oracle calls it like the reference does)."""
    import json
    try:
        data = json.loads(bboxes_json)
        boxes = []
        if isinstance(data, list):
            for it in data:
                if isinstance(it, dict) and "bbox_2d" in it and len(it["bbox_2d"]) == 4:
                    boxes.append(it["bbox_2d"])
    except (json.JSONDecodeError, TypeError):
        boxes = []
    out = []
    h0, w0 = images[0].shape[:2] if images else (0, 0)
    for img in images:
        if img.shape[:2] == (h0, w0) or mask is None:
            out.append(render_overlay(img, mask, boxes))
            continue
        from PIL import Image
        m0 = resize_nearest((np.asarray(mask) > 0).astype(np.uint8), h0, w0) > 0
        ov = np.zeros((h0, w0, 4), dtype=np.uint8)
        ov[m0] = (255, 0, 0, 102)
        base = Image.fromarray(render_overlay(img, None, boxes)).convert("RGBA")
        over = Image.fromarray(ov, "RGBA").resize(base.size, Image.Resampling.LANCZOS)
        out.append(np.array(Image.alpha_composite(base, over).convert("RGB")))
    return out
