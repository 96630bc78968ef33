"""Device raster tail (K19-K22): thin wrappers over the C ABI, torch uint8 CUDA tensors in and out.
Reference behaviour: /root/reference/roll/distributed/strategy/seg_strategy.py:37-69 and
/root/reference/roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:45-58, 383-452."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


def _s(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def mask_union_(acc: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """acc = logical_or(acc, mask).astype(uint8), in place."""
    assert acc.dtype == torch.uint8 and mask.dtype == torch.uint8 and acc.is_cuda and acc.numel() == mask.numel()
    mask = mask.contiguous()
    L.check(L.load().sr_mask_union(C.c_void_p(acc.data_ptr()), C.c_void_p(mask.data_ptr()), acc.numel(), _s(acc)), None, "sr_mask_union")
    return acc


def resize_nearest(src: torch.Tensor, H: int, W: int) -> torch.Tensor:
    assert src.dtype == torch.uint8 and src.is_cuda and src.dim() == 2
    src = src.contiguous()
    dst = torch.empty(H, W, dtype=torch.uint8, device=src.device)
    L.check(L.load().sr_resize_nearest_u8(C.c_void_p(src.data_ptr()), src.shape[0], src.shape[1], C.c_void_p(dst.data_ptr()), H, W, _s(src)),
            None, "sr_resize_nearest_u8")
    return dst


def iou_counts(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """int64 [2] = (intersection, union) on the device."""
    assert pred.dtype == torch.uint8 and gt.dtype == torch.uint8 and pred.numel() == gt.numel()
    pred, gt = pred.contiguous(), gt.contiguous()
    out = torch.empty(2, dtype=torch.int64, device=pred.device)
    L.check(L.load().sr_iou_counts(C.c_void_p(pred.data_ptr()), C.c_void_p(gt.data_ptr()), pred.numel(), C.c_void_p(out.data_ptr()), _s(pred)),
            None, "sr_iou_counts")
    return out


def iou_counts_batched(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """(intersection, union) pixel counts of n (prediction, ground truth) pairs, uint8 [n, h, w] each, in ONE launch -> int64 [n, 2]."""
    assert pred.dtype == torch.uint8 and gt.dtype == torch.uint8 and pred.is_cuda and pred.shape == gt.shape and pred.dim() == 3
    pred, gt = pred.contiguous(), gt.contiguous()
    n = int(pred.shape[0])
    out = torch.empty(n, 2, dtype=torch.int64, device=pred.device)
    L.check(L.load().sr_iou_counts_batched(C.c_void_p(pred.data_ptr()), C.c_void_p(gt.data_ptr()), pred[0].numel() if n else 0, n, C.c_void_p(out.data_ptr()), _s(pred)),
            None, "sr_iou_counts_batched")
    return out


def compute_giou(pred: torch.Tensor, gt: torch.Tensor) -> float:
    i, u = iou_counts(pred, gt).tolist()
    return 1.0 if u == 0 else i / u


def pil_rect(box):
    """One model-generated ``bbox_2d`` -> the 4 ints PIL's ImageDraw.rectangle([(b0, b1), (b2, b3)]) would draw, or None when
    PIL would raise (the reference then skips the box, rlvr_socioseg_vlm_pipeline_infer.py:427-433): coordinates must be
    plain numbers, x1 >= x0 and y1 >= y0 are checked on the float values, then each value is truncated toward zero
    ((int) cast in PIL's _draw_rectangle).  Non-finite / beyond-int32 values map to -2^30 (PIL's cast yields INT_MIN; both
    lie off every image and draw the same pixels)."""
    try:
        if len(box) != 4:
            return None
        vals = []
        for v in box:
            if isinstance(v, (bool, int, float)):
                vals.append(float(v))
            else:
                return None
    except (TypeError, OverflowError):
        return None
    if vals[2] < vals[0] or vals[3] < vals[1]:
        return None
    return [int(v) if (v == v and -2147483648.0 < v < 2147483648.0) else -(1 << 30) for v in vals]


def render_overlay_(img_rgb: torch.Tensor, mask: torch.Tensor | None, boxes) -> torch.Tensor:
    """In place on a uint8 [H,W,3] CUDA image: PIL's width-2 blue outline for every drawable box of ``boxes`` (malformed
    boxes are skipped like the reference's try/except does), then the 40 % red overlay where mask>0."""
    assert img_rgb.dtype == torch.uint8 and img_rgb.is_cuda and img_rgb.is_contiguous()
    rects = [r for r in (pil_rect(b) for b in boxes) if r is not None]
    bx = torch.as_tensor(rects, dtype=torch.int32).reshape(-1, 4).to(img_rgb.device)
    h, w = img_rgb.shape[:2]
    if mask is not None:
        mask = mask.contiguous()
    L.check(L.load().sr_render_overlay(C.c_void_p(img_rgb.data_ptr()), h, w, C.c_void_p(mask.data_ptr()) if mask is not None else None,
                                       mask.shape[0] if mask is not None else 0, mask.shape[1] if mask is not None else 0,
                                       C.c_void_p(bx.data_ptr()) if bx.numel() else None, bx.shape[0], _s(img_rgb)), None, "sr_render_overlay")
    return img_rgb
