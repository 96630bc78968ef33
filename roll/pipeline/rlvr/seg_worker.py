"""``SegWorker`` -- text -> SAM prompts -> mask (reference: roll/pipeline/rlvr/seg_worker.py:199-259, 787-930; SURVEY.md
rows A10 / A11).  Decodes the LM responses (special tokens skipped), parses ``<answer>[...]</answer>`` into per-object
``{box, point_coords, point_labels}`` dicts, and hands them to the ``seg_infer`` strategy, whose raster half (union,
nearest 756 -> 768) runs on the device."""
from __future__ import annotations

from typing import Any, Dict, List

import numpy as np
import torch

from roll.distributed.scheduler.protocol import DataProto
from roll.pipeline.base_worker import Worker
from socioreasoner_amd.hostops import parse_points_text_from_content, parse_visual_prompt_from_json_s2  # noqa: F401


def build_sam_prompts(response: str) -> List[Dict[str, Any]]:
    """One LM response -> the list of SAM2 ``predict`` keyword dicts (reference :787-826)."""
    out = []
    for obj in parse_visual_prompt_from_json_s2(response):
        d: Dict[str, Any] = {}
        try:
            if obj.get("box") and len(obj["box"]) == 4:
                d["box"] = np.array(obj["box"])
            if obj.get("points"):
                pc, pl = np.array(obj["points"]), np.array(obj["labels"])
                if pc.ndim == 2 and pl.ndim == 1 and pc.shape[0] == pl.shape[0] and pc.shape[1] == 2:
                    d["point_coords"], d["point_labels"] = pc, pl
        except Exception:  # noqa: BLE001  (malformed objects are dropped, as in the reference)
            pass
        if d:
            out.append(d)
    return out


class SegWorker(Worker):
    def _segment(self, data: DataProto, response_key: str) -> DataProto:
        texts = self.tokenizer.batch_decode(data.batch[response_key], skip_special_tokens=True)
        prompts = np.empty(len(texts), dtype=object)
        for i, t in enumerate(texts):
            prompts[i] = build_sam_prompts(t)
        data.non_tensor_batch["visual_prompt"] = prompts
        out = self.strategy.segment(batch=data)
        data.non_tensor_batch["mask"] = out["mask"]
        rt = np.empty(len(texts), dtype=object)
        rt[:] = texts
        data.non_tensor_batch["response_text"] = rt
        data.meta_info = {"metrics": {}}
        return data

    def prefetch_images(self, images, chunk=None) -> None:
        """hint: these images will be segmented soon (the strategy may start SAM2's image encoder now; no reference counterpart).  chunk: images per
        encoder pass of the background work (a segment call that arrives meanwhile waits for at most one pass)"""
        if hasattr(self.strategy, "prefetch"):
            self.strategy.prefetch(images, chunk=chunk)

    def wait_prefetch(self) -> None:
        """returns when the background work started by prefetch_images has ended (raises what it raised)"""
        if hasattr(self.strategy, "wait_prefetch"):
            self.strategy.wait_prefetch()

    @torch.no_grad()
    def segment_v4_map(self, data: DataProto) -> DataProto:
        return self._segment(data, "map_responses")

    @torch.no_grad()
    def segment_v4_sat(self, data: DataProto) -> DataProto:
        return self._segment(data, "responses")
