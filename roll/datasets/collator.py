"""``DataCollatorWithPaddingForMultiSeg`` -- the batch builder in front of the hot path (reference:
roll/datasets/collator.py:414-564).

Per sample it runs the processor on (images=[map, sat], text=stage-1 prompt), keeps the engine payload
``{"prompt_token_ids", "multi_modal_data": {"image": [...]}}`` the strategy's ``generate`` consumes, pads ids / mask to
``max_length`` on the tokenizer's padding side (left), and asks ``extra_data_provider`` for the mRoPE ``position_ids``
stored as ``(B, 3, S)``.  Output keys and shapes are the reference's: ``map_input_ids``, ``map_attention_mask``,
``map_position_ids`` tensors; every other field an object array of length B.  Works with the HF processor / tokenizer
of a real checkpoint or with socioreasoner_amd.textproc's offline stand-ins.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import torch


def collate_fn_to_dict_list(data_list: List[Dict[str, Any]]) -> Dict[str, Any]:
    """list of dicts -> dict: tensors concatenated along dim 0, everything else an object array (reference :13-37)."""
    tensors: Dict[str, List[torch.Tensor]] = {}
    others: Dict[str, List[Any]] = {}
    for d in data_list:
        for k, v in d.items():
            (tensors if isinstance(v, torch.Tensor) else others).setdefault(k, []).append(v)
    out: Dict[str, Any] = {k: torch.cat(v, dim=0) for k, v in tensors.items()}
    for k, v in others.items():
        out[k] = _object_array(v)
    return out


def _object_array(values: List[Any]) -> np.ndarray:
    arr = np.empty([len(values)], dtype=object)
    arr[:] = values
    return arr


@dataclass
class DataCollatorWithPaddingForMultiSeg:
    tokenizer: Any = None
    processor: Any = None
    extra_data_provider: Optional[Callable] = None
    prompt_map_key: Optional[str] = "prompt_map"
    question_key: Optional[str] = "question"
    image_key: Optional[str] = None
    map_image_key: Optional[str] = "image_map"
    id_key: Optional[str] = "id"
    gt_mask_key: Optional[str] = "gt_mask"
    gt_point_key: Optional[str] = None
    seg_image_key: Optional[str] = "seg_image"
    gt_object_key: Optional[str] = None
    gt_center_key: Optional[str] = None
    gt_bbox_key: Optional[str] = None
    image_flag_key: Optional[str] = "image_flag"
    padding: Any = True
    max_length: Optional[int] = None
    pad_to_multiple_of: Optional[int] = None
    padded_keys: List[str] = field(default_factory=lambda: ["input_ids", "attention_mask", "labels"])
    return_tensors: str = "pt"

    def _passthrough_keys(self) -> List[str]:
        keys = [self.question_key, self.gt_mask_key, self.gt_object_key, self.gt_point_key, self.gt_center_key,
                self.seg_image_key, self.map_image_key, self.gt_bbox_key, self.image_key, self.id_key]
        return [k for k in keys if k]

    def __call__(self, features: List[Dict[str, Any]]) -> Dict[str, Any]:
        assert self.tokenizer is not None and self.processor is not None
        padded: Dict[str, List] = {}
        loose: Dict[str, List] = {}
        mm_keys = set()
        for feat in features:
            has_image = bool(self.image_key) and (not self.image_flag_key or bool(feat[self.image_flag_key]))
            text = feat[self.prompt_map_key] if self.prompt_map_key else None
            enc = self.processor(images=feat[self.image_key] if has_image else None, text=text)
            enc.pop("prompt_map", None)
            for k in self.padded_keys:
                if k in enc:
                    padded.setdefault(k, []).append(enc.pop(k)[0])
            mm_keys |= set(enc.keys())
            enc.convert_to_tensors(tensor_type=self.return_tensors)
            if self.image_key:
                loose.setdefault("multi_modal_map_inputs", []).append(dict(enc))
                payload = {"prompt_token_ids": self.tokenizer.encode(text, add_special_tokens=False) if text else []}
                if has_image:
                    im = feat[self.image_key]
                    payload["multi_modal_data"] = {"image": im if isinstance(im, list) else [im]}
                loose.setdefault("multi_modal_map_data", []).append(payload)
            for k in self._passthrough_keys():
                loose.setdefault(k, []).append(feat[k])

        pad_fn = getattr(self.tokenizer, "pad")
        pb = pad_fn(padded, padding=self.padding, max_length=self.max_length, pad_to_multiple_of=self.pad_to_multiple_of,
                    return_tensors=self.return_tensors)
        batch: Dict[str, Any] = {"map_input_ids": pb["input_ids"], "map_attention_mask": pb["attention_mask"]}
        batch.update(loose)

        if self.extra_data_provider:
            kwargs = {"input_ids": batch["map_input_ids"], "attention_mask": batch["map_attention_mask"]}
            if "image_grid_thw" in mm_keys:
                grids = [d["image_grid_thw"] for d in batch["multi_modal_map_inputs"] if "image_grid_thw" in d]
                kwargs["image_grid_thw"] = torch.cat(grids, dim=0) if grids else None
            extra = self.extra_data_provider(**kwargs)
            batch["map_position_ids"] = extra.pop("position_ids")
            batch.update(extra)

        B = batch["map_input_ids"].shape[0]
        for k, v in list(batch.items()):
            if isinstance(v, (torch.Tensor, np.ndarray)):
                assert v.shape[0] == B, k
            else:
                assert len(v) == B, k
                batch[k] = _object_array(v)
        return batch
