"""Synthetic workload of SURVEY.md section 8(D): there is no dataset, tokenizer or checkpoint offline.
One "tile" = one 448x448 RGB uint8 crop + a 448-token prompt (96 text + <vision_start> + 256 <image_pad> +
<vision_end> + 94 text) + 4 synthetic 756x756 object masks and a 768x768 ground truth for the raster tail."""
from __future__ import annotations

import numpy as np

from .config import ModelGeometry


def tile_pixels(i: int, h: int = 448, w: int = 448) -> np.ndarray:
    return np.random.default_rng(1000 + i).integers(0, 256, (h, w, 3), dtype=np.uint8)


def tile_prompt(g: ModelGeometry, i: int, grid_thw, n_pre: int = 96, n_post: int = 94, n_images: int = 1) -> np.ndarray:
    """n_images = 1: the bench tile (448 tokens).  n_images = 2: the reference-faithful (map, satellite) pair the
    pipeline feeds in both stages (/root/reference/roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:61-124)."""
    t, h, w = grid_thw
    T = t * h * w // (g.vision.spatial_merge_size ** 2)
    rng = np.random.default_rng(2000 + i)
    hi = min(g.image_token_id, g.text.vocab_size) - 13    # text ids stay below the special-token block
    pre, post = rng.integers(0, hi, n_pre), rng.integers(0, hi, n_post)
    img = np.concatenate([[g.vision_start_token_id], np.full(T, g.image_token_id), [g.vision_end_token_id]])
    return np.concatenate([pre] + [img] * n_images + [post]).astype(np.int64)


def tile_masks(i: int, n_obj: int = 4, size: int = 756, gt_size: int = 768):
    rng = np.random.default_rng(3000 + i)
    masks = np.zeros((n_obj, size, size), dtype=np.uint8)
    for j in range(n_obj):
        x0, y0 = rng.integers(0, size - 160, 2)
        ww, hh = rng.integers(20, 150, 2)
        masks[j, y0:y0 + hh, x0:x0 + ww] = 1
    gt = np.zeros((gt_size, gt_size), dtype=np.uint8)
    x0, y0 = rng.integers(0, gt_size - 300, 2)
    gt[y0:y0 + 280, x0:x0 + 280] = 255
    return masks, gt
