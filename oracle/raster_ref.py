"""ctypes loader for oracle/raster_ref.c.  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libraster_ref.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "raster_ref.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = ctypes.CDLL(so)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def mask_union(masks):
    acc = np.zeros(masks[0].shape, dtype=np.uint8)
    for m in masks:
        m = np.ascontiguousarray(m, dtype=np.uint8)
        lib().ref_mask_union(_p(acc), _p(m), ctypes.c_size_t(acc.size))
    return acc


def resize_nearest(src, H, W):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty((H, W), dtype=np.uint8)
    lib().ref_resize_nearest_u8(_p(src), src.shape[0], src.shape[1], _p(dst), H, W)
    return dst


def iou_counts(pred, gt):
    pred = np.ascontiguousarray(pred, dtype=np.uint8)
    gt = np.ascontiguousarray(gt, dtype=np.uint8)
    out = np.zeros(2, dtype=np.int64)
    lib().ref_iou_counts(_p(pred), _p(gt), ctypes.c_size_t(pred.size), _p(out))
    return int(out[0]), int(out[1])


def render_overlay(img_rgb, mask, boxes):
    from .host_ref import pil_box
    img = np.ascontiguousarray(img_rgb, dtype=np.uint8).copy()
    boxes = [ib for ib in (pil_box(b if not isinstance(b, np.ndarray) else b.tolist()) for b in boxes) if ib is not None]
    boxes = np.ascontiguousarray(np.asarray(boxes, dtype=np.int32).reshape(-1, 4))
    if mask is None:
        lib().ref_render_overlay(_p(img), img.shape[0], img.shape[1], None, 0, 0, _p(boxes), len(boxes))
    else:
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        lib().ref_render_overlay(_p(img), img.shape[0], img.shape[1], _p(mask), mask.shape[0], mask.shape[1],
                                 _p(boxes), len(boxes))
    return img


def normalize_lut():
    lut = np.zeros(768, dtype=np.float32)
    lib().ref_normalize_lut(_p(lut))
    return lut.reshape(3, 256)
