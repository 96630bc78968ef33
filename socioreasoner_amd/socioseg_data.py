"""SocioSeg samples for the infer pipeline (reference: roll/datasets/dataset.py:49-119 folder layout;
roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:146-184 ground-truth helpers).

A sample is ``{"id", "problem", "map_image", "sat_image", "mask_label"}`` with PIL images (or paths).  The reference
pulls ``vvangfaye/SocioSeg`` from the hub; offline the pipeline reads the same folder layout from disk or falls back to
synthetic tiles (BASELINE.json: data = synthetic).  cv2 is not available here: connected components / bounding boxes
are restated with scipy.ndimage (8-connectivity like the reference's connectedComponentsWithStats; the box filter is
component pixel area > 10, the reference filters on polygon contourArea > 10 -- differs only for thread-like blobs).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List

import numpy as np

from socioreasoner_amd import synthetic


def _pil():
    from PIL import Image
    return Image


def load_image(x):
    Image = _pil()
    if isinstance(x, str):
        with Image.open(x) as im:
            return im.copy()
    if isinstance(x, np.ndarray):
        return Image.fromarray(x)
    return x


def load_socioseg_folder(data_dir: str, split: str = "test") -> List[Dict]:
    """<data_dir>/<split>/<id>/{map.png, sat.png, mask.png, question.json}; incomplete samples are skipped."""
    root = os.path.join(data_dir, split)
    out = []
    for sid in sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))):
        p = os.path.join(root, sid)
        files = {k: os.path.join(p, k) for k in ("question.json", "map.png", "sat.png", "mask.png")}
        if not all(os.path.exists(f) for f in files.values()):
            continue
        try:
            problem = json.load(open(files["question.json"], encoding="utf-8")).get("problem", "")
        except Exception:  # noqa: BLE001
            continue
        out.append({"id": sid, "problem": problem, "map_image": files["map.png"], "sat_image": files["sat.png"],
                    "mask_label": files["mask.png"]})
    return out


def write_socioseg_folder(samples: List[Dict], data_dir: str, split: str = "test") -> None:
    for s in samples:
        p = os.path.join(data_dir, split, s["id"])
        os.makedirs(p, exist_ok=True)
        load_image(s["map_image"]).save(os.path.join(p, "map.png"))
        load_image(s["sat_image"]).save(os.path.join(p, "sat.png"))
        load_image(s["mask_label"]).save(os.path.join(p, "mask.png"))
        json.dump({"problem": s["problem"]}, open(os.path.join(p, "question.json"), "w", encoding="utf-8"))


_PROBLEMS = ["residential area", "the commercial district near the river", "school", "parks and green space", "industrial zone"]


def synthetic_socioseg(n: int, first: int = 0, size: int = 448) -> List[Dict]:
    """Synthetic tiles in SocioSeg form: seeded uint8 noise for map / satellite, ground truth = the
    768 x 768 rectangle mask of synthetic.tile_masks."""
    Image = _pil()
    out = []
    for i in range(first, first + n):
        _, gt = synthetic.tile_masks(i)
        out.append({"id": f"synthetic_{i:06d}", "problem": _PROBLEMS[i % len(_PROBLEMS)],
                    "map_image": Image.fromarray(synthetic.tile_pixels(10_000 + i, size, size)),
                    "sat_image": Image.fromarray(synthetic.tile_pixels(i, size, size)),
                    "mask_label": Image.fromarray((gt > 0).astype(np.uint8) * 255, mode="L")})
    return out


def _binary(img) -> np.ndarray:
    a = np.asarray(img)
    if a.ndim == 3:     # cv2.COLOR_RGB2GRAY then threshold(>0): any channel that survives the luma rounding
        a = (0.299 * a[..., 0] + 0.587 * a[..., 1] + 0.114 * a[..., 2] + 0.5).astype(np.int64)
    return a > 0


def count_components(image_list) -> List[int]:
    from scipy import ndimage
    return [int(ndimage.label(_binary(im), structure=np.ones((3, 3)))[1]) for im in image_list]


def get_bboxes(image_list) -> List[str]:
    """JSON list of {"bbox_2d": [x0, y0, x1, y1]} (exclusive max, like cv2.boundingRect's x + w) per image."""
    from scipy import ndimage
    out = []
    for im in image_list:
        lab, n = ndimage.label(_binary(im), structure=np.ones((3, 3)))
        boxes = []
        for k, sl in enumerate(ndimage.find_objects(lab), start=1):
            if sl is None or int((lab[sl] == k).sum()) <= 10:
                continue
            boxes.append({"bbox_2d": [int(sl[1].start), int(sl[0].start), int(sl[1].stop), int(sl[0].stop)]})
        out.append(json.dumps(boxes))
    return out


class SyntheticSamPredictor:
    """Stand-in for SAM2ImagePredictor (set_image / predict) so that the raster tail after SAM2 runs offline: three
    candidate masks per prompt -- the filled box, the box shrunk by an eighth per side, the box united with radius-20
    discs around the positive points -- with fixed scores (the pipeline keeps the arg-max candidate,
    seg_strategy.py:58-60).  NOT a segmentation model."""

    def set_image(self, image):
        self.w, self.h = image.size if hasattr(image, "size") and not isinstance(image, np.ndarray) else image.shape[1::-1]

    def predict(self, box=None, point_coords=None, point_labels=None, **_):
        h, w = self.h, self.w
        m = np.zeros((3, h, w), dtype=bool)
        scores = np.array([0.2, 0.1, 0.05], dtype=np.float32)
        if box is not None:
            x0, y0, x1, y1 = [int(round(float(v))) for v in np.asarray(box).reshape(-1)[:4]]
            x0, x1 = sorted((min(max(x0, 0), w), min(max(x1, 0), w)))
            y0, y1 = sorted((min(max(y0, 0), h), min(max(y1, 0), h)))
            m[0, y0:y1, x0:x1] = True
            qx, qy = (x1 - x0) // 8, (y1 - y0) // 8
            m[1, y0 + qy:y1 - qy, x0 + qx:x1 - qx] = True
            scores[:2] = (0.9, 0.6)
        if point_coords is not None and len(point_coords):
            yy, xx = np.mgrid[0:h, 0:w]
            m[2] = m[0]
            for (px, py), lab in zip(np.asarray(point_coords), np.asarray(point_labels)):
                if int(lab) == 1:
                    m[2] |= (xx - float(px)) ** 2 + (yy - float(py)) ** 2 <= 20 ** 2
            scores[2] = 0.95
        return m, scores, None
