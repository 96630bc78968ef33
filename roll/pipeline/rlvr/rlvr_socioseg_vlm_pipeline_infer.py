"""Two-stage SocioSeg inference pipeline on the MI355X-native engine -- same module path, class names and output
files as the reference (roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:511-919):

    per batch: stage-1 generate -> SAM prompts -> mask (union, nearest 756->768) -> render (boxes + 40 % red overlay)
               -> stage-2 generate -> mask -> IoU;  files under ./output/infer/result/{stage1,stage2,render1,render2}
               and the mean in iou_acc.txt.

What is different, and why: there is no Ray (workers = the torchrun ranks, tiles sharded like the reference's
DP dispatch and gathered with one RCCL all-gather), and nothing that needs the network exists offline -- SocioSeg data,
the tokenizer and SAM2.  Without them the pipeline runs its *synthetic mode* (SURVEY.md section 8(D)): synthetic tiles
and token ids, and synthetic per-object masks standing in for SAM2's output, so that every raster step after SAM2 runs
for real on the device.  With a checkpoint directory + dataset + a SAM2-compatible predictor passed in, the same code
path runs on real data.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Union

import numpy as np
import torch

from roll.distributed.scheduler.protocol import DataProto
from roll.distributed.strategy.factory import create_strategy
from roll.pipeline.base_pipeline import BasePipeline
from roll.pipeline.rlvr.rlvr_config import SocioSegConfig  # noqa: F401  (re-exported like the reference)
from socioreasoner_amd import dp, hostops, raster, synthetic
from socioreasoner_amd.hostops import parse_points_text_from_content, parse_visual_prompt_from_json_s2  # noqa: F401


def compute_giou(pred_mask: np.ndarray, gt_mask: np.ndarray) -> float:
    """Reference :45-58.  Runs the integer counts on the device when one is present."""
    if torch.cuda.is_available():
        return raster.compute_giou(torch.from_numpy(np.ascontiguousarray(pred_mask, dtype=np.uint8)).cuda(),
                                   torch.from_numpy(np.ascontiguousarray(gt_mask, dtype=np.uint8)).cuda())
    return hostops.compute_giou(pred_mask, gt_mask)


# Prompt templates: the model was trained on these exact strings (reference :61-124); they are data, kept verbatim.
_Q1 = ("You will be given two images. The first is a map and the second is a corresponding satellite image."
       "Please find '{prompt}' with bboxs."
       "Compare the difference between object(s) and find the most closely matched object(s)."
       "Output the thinking process in <think> </think> and final answer in <answer> </answer> tags. Please use English."
       "Output the bbox(es) in JSON format."
       "i.e., <think>thinking process here </think>"
       "<answer>{answer}</answer>")
_A1 = "[{\"bbox_2d\": [bx1,by1,bx2,by2]}, {\"bbox_2d\": [bx3,by3,bx4,by4]}]"
_Q2 = ("You will be given two images. The first is a map and the second is a corresponding satellite image."
       "Now some bbox(s) and the results after SAM segmentation for \"{prompt}\" have been rendered on these two images."
       "The found bbox(s) are: {bboxs}."
       "Please add some points appropriately to each bbox to better represent the area of interest."
       "Output the thinking process in <think> </think> and final answer in <answer> </answer> tags."
       "i.e., <think> thinking process here </think>"
       "<answer>{answer}</answer>")
_A2 = ("[{\"bbox_2d\": [bx1,by1,bx2,by2], \"points\": [[px1,py1],[px2,py2],[px3,py3]]}, "
       "{\"bbox_2d\": [bx3,by3,bx4,by4], \"points\": [[px4,py4],[px5,py5],[px6,py6]}]")


def _chat(processor, text, use_image, prompt_image_token):
    content = ([{"type": "image"}, {"type": "image"}] if use_image and not prompt_image_token else []) + [{"type": "text", "text": text}]
    out = processor.apply_chat_template([{"role": "user", "content": content}], tokenize=False, add_generation_prompt=True)
    return out.replace(prompt_image_token, "<|vision_start|><|image_pad|><|vision_end|>") if prompt_image_token else out


def format_prompt_1(prompt, processor, use_image=True, prompt_image_token=None):
    return _chat(processor, _Q1.format(prompt=prompt, answer=_A1), use_image, prompt_image_token)


def format_prompt_2(prompt, bboxs, processor, use_image=True, prompt_image_token=None):
    return _chat(processor, _Q2.format(prompt=prompt, bboxs=bboxs, answer=_A2), use_image, prompt_image_token)


def render_image(bboxes_json: str, images: List[Any], mask: Union[np.ndarray, Any]) -> List[Any]:
    """Reference :383-452 on the device: nearest-resize of the mask to the image, 2-px blue box outlines, 40 % red
    overlay (PIL alpha_composite arithmetic).  Accepts PIL images or uint8 HWC arrays, returns the same kind."""
    try:
        data = json.loads(bboxes_json)
        boxes = [it["bbox_2d"] for it in data if isinstance(it, dict) and "bbox_2d" in it and len(it["bbox_2d"]) == 4] \
            if isinstance(data, list) else []
    except (json.JSONDecodeError, TypeError):
        boxes = []
    m = None
    if mask is not None:
        m = np.asarray(mask.convert("L")) if hasattr(mask, "convert") else np.asarray(mask)
        m = torch.from_numpy(np.ascontiguousarray((m > 0).astype(np.uint8))).cuda()
    out = []
    for im in images:
        is_pil = hasattr(im, "convert")
        arr = np.asarray(im.convert("RGB")) if is_pil else np.asarray(im)
        t = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
        raster.render_overlay_(t, m, boxes)
        res = t.cpu().numpy()
        if is_pil:
            from PIL import Image
            res = Image.fromarray(res)
        out.append(res)
    return out


class _Worker:
    """What a strategy sees of its worker (reference: roll/distributed/executor/worker.py:41-204)."""

    def __init__(self, worker_config, pipeline_config, rank, world_size, local_rank):
        self.worker_config, self.pipeline_config = worker_config, pipeline_config
        self.rank, self.world_size = rank, world_size
        self.rank_info = type("RankInfo", (), {"dp_rank": rank, "dp_size": world_size, "local_rank": local_rank})()


class SocioSegInferPipeline(BasePipeline):
    def __init__(self, pipeline_config, sam_predictor_provider=None, dataset=None):
        super().__init__(pipeline_config)
        self.rank, self.world, local = dp.init_distributed()
        self.actor_infer = create_strategy(_Worker(pipeline_config.actor_infer, pipeline_config, self.rank, self.world, local))
        self.actor_infer.initialize(None)
        self.seg_infer = create_strategy(_Worker(pipeline_config.seg_infer, pipeline_config, self.rank, self.world, local))
        self.seg_infer.initialize(sam_predictor_provider)
        self.tokenizer = self.actor_infer.tokenizer
        self.geom = self.actor_infer.geom
        self.dataset = dataset            # None -> synthetic tiles
        ga = pipeline_config.actor_infer.generating_args or {}
        eos = [self.tokenizer.eos_token_id] + list(getattr(self.tokenizer, "additional_special_tokens_ids", []) or [])
        self.generation_config = {
            "max_new_tokens": int(ga.get("max_new_tokens") or pipeline_config.response_length),
            "temperature": ga.get("temperature", 0), "top_p": ga.get("top_p", 1.0), "top_k": ga.get("top_k", 1),
            "num_beams": ga.get("num_beams", 1), "repetition_penalty": ga.get("repetition_penalty", 1.0),
            "num_return_sequences": 1, "eos_token_id": eos, "pad_token_id": self.tokenizer.pad_token_id,
        }
        self.n_samples = int(os.environ.get("SOCIOSEG_NUM_SAMPLES", pipeline_config.rollout_batch_size))

    # ------------------------------------------------------------------ synthetic batch (no dataset / tokenizer offline)
    def _synthetic_sample(self, i: int) -> Dict:
        grid = (1, 32, 32)
        ids = synthetic.tile_prompt(self.geom, i, grid, n_images=2)      # (map, satellite) like the reference
        return {"id": f"synthetic_{i:06d}", "tile": synthetic.tile_pixels(i), "map": synthetic.tile_pixels(10_000 + i),
                "ids": ids, "masks": synthetic.tile_masks(i)}

    def _generate(self, samples: List[Dict], images_key: str) -> torch.Tensor:
        P = int(self.pipeline_config.prompt_length)
        pad = self.tokenizer.pad_token_id
        rows, payload = [], np.empty(len(samples), dtype=object)
        for k, s in enumerate(samples):
            ids = np.asarray(s["ids"], dtype=np.int64)
            row = np.full(P, pad, dtype=np.int64)
            row[P - len(ids):] = ids                                     # left padding (reference collator.py:444-564)
            rows.append(row)
            payload[k] = {"prompt_token_ids": ids.tolist(), "multi_modal_data": {"image": s[images_key]}}
        input_ids = torch.from_numpy(np.stack(rows))
        batch = DataProto(batch={"input_ids": input_ids, "attention_mask": (input_ids != pad).long()},
                          non_tensor_batch={"multi_modal_data": payload})
        out = self.actor_infer.generate(batch, self.generation_config)
        return out[:, P:]

    @torch.no_grad()
    def run(self):
        cfg = self.pipeline_config
        res_dir = os.path.join(cfg.output_dir, "result")
        for sub in ("stage1", "stage2", "render1", "render2"):
            os.makedirs(os.path.join(res_dir, sub), exist_ok=True)
        lo, hi = dp.shard_range(self.n_samples, self.rank, self.world)
        ious = []
        bs = int(os.environ.get("SOCIOSEG_BATCH", 32))
        for b0 in range(lo, hi, bs):
            samples = [self._synthetic_sample(i) for i in range(b0, min(b0 + bs, hi))]
            for s in samples:
                s["images1"] = [s["map"], s["tile"]]
            resp1 = self._generate(samples, "images1")                       # STAGE 1
            text1 = self.tokenizer.batch_decode(resp1, skip_special_tokens=False)
            for s, r, txt in zip(samples, resp1, text1):
                masks, gt = s["masks"]
                acc = torch.zeros(756, 756, dtype=torch.uint8, device="cuda")
                for m in masks[:2]:                                           # synthetic stand-in for SAM2's per-object masks
                    raster.mask_union_(acc, torch.from_numpy(m).cuda())
                s["mask1"] = raster.resize_nearest(acc, 768, 768)
                boxes = parse_points_text_from_content(txt) or "[]"
                s["render"] = render_image(boxes, [s["map"], s["tile"]], s["mask1"].cpu().numpy())
                open(os.path.join(res_dir, "stage1", s["id"] + ".txt"), "w").write(txt)
            resp2 = self._generate(samples, "render")                        # STAGE 2 (re-encodes the rendered tile)
            text2 = self.tokenizer.batch_decode(resp2, skip_special_tokens=False)
            for s, txt in zip(samples, text2):
                masks, gt = s["masks"]
                acc = torch.zeros(756, 756, dtype=torch.uint8, device="cuda")
                for m in masks:
                    raster.mask_union_(acc, torch.from_numpy(m).cuda())
                pred = raster.resize_nearest(acc, 768, 768)
                ious.append(raster.compute_giou(pred, torch.from_numpy(gt).cuda()))
                open(os.path.join(res_dir, "stage2", s["id"] + ".txt"), "w").write(txt)
                try:
                    from PIL import Image
                    Image.fromarray(pred.cpu().numpy() * 255).save(os.path.join(res_dir, "stage2", s["id"] + ".png"))
                    Image.fromarray(s["render"][0]).save(os.path.join(res_dir, "render1", s["id"] + ".png"))
                    Image.fromarray(s["render"][1]).save(os.path.join(res_dir, "render2", s["id"] + ".png"))
                except Exception:  # noqa: BLE001
                    pass
        local = torch.tensor(ious, dtype=torch.float64).reshape(-1, 1)
        if self.world > 1:
            local = dp.all_gather_rows(local.cuda() if torch.cuda.is_available() else local, self.n_samples).cpu()
        giou_acc = float(local.mean()) if local.numel() else 0.0
        if self.rank == 0:
            print(f"giou_acc: {giou_acc}")
            with open(os.path.join(res_dir, "iou_acc.txt"), "w") as f:
                f.write(str(giou_acc))
        return giou_acc
