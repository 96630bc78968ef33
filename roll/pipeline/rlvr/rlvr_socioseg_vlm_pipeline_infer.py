"""Two-stage SocioSeg inference pipeline on the MI355X-native engine -- same module path, class names and output
files as the reference (roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:511-919):

    per batch: stage-1 generate -> SAM prompts -> mask (union, nearest 756->768) -> render (boxes + 40 % red overlay)
               -> stage-2 generate -> mask -> IoU;  files under ./output/infer/result/{stage1,stage2,render1,render2}
               and the mean in iou_acc.txt.

What is different, and why: there is no Ray (workers = the torchrun ranks; samples are sharded like the reference's DP
dispatch and the per-sample IoUs gathered with one RCCL all-gather), and nothing that needs the network exists offline
-- SocioSeg data, the tokenizer and SAM2.  Each has a stand-in behind the SAME interface, so one code path serves both:
  data      : a SocioSeg folder (roll/datasets/dataset.py layout) via data_args.dataset_dir, else synthetic tiles;
  processor : the checkpoint's HF processor when `pretrain` is a directory, else textproc.SyntheticProcessor
              (byte-level tokenizer with the Qwen special-token ids);
  SAM2      : `sam_predictor_provider` (anything with set_image / predict), else roll.models.model_providers.sam2_seg_model_provider (the
              MI355X SAM2) when seg_infer.model_args names a model, else SyntheticSamPredictor, which turns the
              parsed boxes / points into masks so that every raster step after SAM2 runs for real on the device.
The batch keys, the order of operations and the files written follow the reference's `run()` step by step.
"""
from __future__ import annotations

import contextlib
import json
import os
from collections import defaultdict
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

from roll.datasets.collator import DataCollatorWithPaddingForMultiSeg
from roll.distributed.scheduler.generate_scheduler import GenerateScheduler
from roll.distributed.scheduler.protocol import DataProto
from roll.pipeline.base_pipeline import BasePipeline
from roll.pipeline.base_worker import ActorWorker, Worker
from roll.pipeline.rlvr.rlvr_config import SocioSegConfig  # noqa: F401  (re-exported like the reference)
from roll.pipeline.rlvr.seg_worker import SegWorker
from socioreasoner_amd import dp, hostops, raster, socioseg_data
from socioreasoner_amd.hostops import parse_points_text_from_content, parse_visual_prompt_from_json_s2  # noqa: F401

_Worker = Worker      # earlier name, kept for callers that build a bare worker


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


def render_image(bboxes_json: str, images: List[Any], mask: Union[np.ndarray, Any], keep_on_device: bool = False) -> List[Any]:
    """Reference :383-452 on the device: nearest-resize of the mask to the image, 2-px blue box outlines, 40 % red
    overlay (PIL alpha_composite arithmetic).  Accepts PIL images, uint8 HWC arrays or CUDA tensors; returns the same kind,
    or -- with keep_on_device -- uint8 HWC CUDA tensors that the strategy patchifies without a host round trip (SURVEY
    "next" row N3)."""
    try:
        data = json.loads(bboxes_json)
        boxes = [it["bbox_2d"] for it in data if isinstance(it, dict) and "bbox_2d" in it and len(it["bbox_2d"]) == 4] \
            if isinstance(data, list) else []
    except (json.JSONDecodeError, TypeError):
        boxes = []
    m = None
    if mask is not None:
        if isinstance(mask, torch.Tensor):
            m = (mask > 0).to(torch.uint8).cuda().contiguous()
        else:
            mm = np.asarray(mask.convert("L")) if hasattr(mask, "convert") else np.asarray(mask)
            m = torch.from_numpy(np.ascontiguousarray((mm > 0).astype(np.uint8))).cuda()
    out = []
    first_hw = None
    for im in images:
        is_pil = hasattr(im, "convert")
        if isinstance(im, torch.Tensor):
            t = im.cuda().clone()
        else:
            arr = np.asarray(im.convert("RGB")) if is_pil else np.asarray(im)
            t = torch.from_numpy(np.array(arr, dtype=np.uint8, order="C")).cuda()
        hw = (int(t.shape[0]), int(t.shape[1]))
        if first_hw is None:
            first_hw = hw
        if m is None or hw == first_hw:
            raster.render_overlay_(t, m, boxes)
        else:
            t = _render_other_size(t, m, boxes, first_hw)
        if keep_on_device or isinstance(im, torch.Tensor):
            out.append(t)
            continue
        res = t.cpu().numpy()
        if is_pil:
            from PIL import Image
            res = Image.fromarray(res)
        out.append(res)
    return out


def _render_other_size(t: torch.Tensor, m: torch.Tensor, boxes, first_hw) -> torch.Tensor:
    """The reference builds ONE overlay at the FIRST image's size and, for an image of another size, resamples that RGBA
    overlay with LANCZOS before compositing (reference :436-438).  Rare (map and satellite tiles normally share a size), so
    this branch keeps the box outlines on the device and lets PIL -- the library the reference itself calls -- do the
    resample + composite on the host: identical pixels by construction."""
    from PIL import Image
    raster.render_overlay_(t, None, boxes)
    m0 = raster.resize_nearest(m, first_hw[0], first_hw[1]).cpu().numpy() > 0
    ov = np.zeros((first_hw[0], first_hw[1], 4), dtype=np.uint8)
    ov[m0] = (255, 0, 0, int(255 * 0.4))
    base = Image.fromarray(t.cpu().numpy()).convert("RGBA")
    over = Image.fromarray(ov, "RGBA").resize(base.size, Image.Resampling.LANCZOS)
    return torch.from_numpy(np.array(Image.alpha_composite(base, over).convert("RGB"), dtype=np.uint8, order="C")).cuda()


def process_image(images: List[Any], processor) -> List[Any]:
    """Resize to multiples of patch * merge within [min_pixels, max_pixels] with the processor's resampling
    (reference :126-144; same rule as HF's Qwen2-VL image processor)."""
    ip = processor.image_processor
    factor = ip.patch_size * ip.merge_size if "Qwen" in getattr(ip, "image_processor_type", "Qwen") else 28
    out = []
    for im in images:
        rh, rw = hostops.smart_resize(im.height, im.width, factor=factor, min_pixels=ip.min_pixels, max_pixels=ip.max_pixels)
        out.append(im.resize((rw, rh), resample=ip.resample))
    return out


def encode_function(data_i: Dict[str, List], processor, id_key="id", prompt_key="problem", label_key="mask_label",
                    image_map_key="map_image", image_sat_key="sat_image") -> Dict[str, List]:
    """Raw SocioSeg columns -> the regularised fields of the reference (:186-268): resized [map, sat] pair, stage-1 prompt
    text, ground-truth mask / boxes / object count.  An image that fails to load is replaced by a black one and the
    sample is flagged text-only, as in the reference."""
    from PIL import Image
    n = len(data_i[prompt_key])
    enc = defaultdict(list)
    for i in range(n):
        pair, ok = [], True
        for key in (image_map_key, image_sat_key):
            try:
                pair.append(process_image([socioseg_data.load_image(data_i[key][i]).convert("RGB")], processor)[0])
            except Exception:  # noqa: BLE001
                pair.append(Image.new("RGB", (224, 224)))
                ok = False
        try:
            gt = socioseg_data.load_image(data_i[label_key][i])
            seg = socioseg_data.load_image(data_i[image_sat_key][i]).convert("RGB")
        except Exception:  # noqa: BLE001
            gt, seg = Image.new("RGB", (756, 756)), Image.new("RGB", (756, 756))
        enc["id"].append(data_i[id_key][i] if id_key in data_i else f"id_{i}")
        enc["prompt_map"].append(format_prompt_1(data_i[prompt_key][i], processor, use_image=ok))
        enc["question"].append(data_i[prompt_key][i])
        enc["gt_mask"].append(gt)
        enc["gt_bbox"].append(socioseg_data.get_bboxes([gt])[0])
        enc["gt_object"].append(socioseg_data.count_components([gt])[0])
        enc["image_sat"].append([pair[1]])
        enc["image_map"].append([pair[0]])
        enc["seg_image"].append(seg)
        enc["image"].append(pair)
        enc["image_flag"].append(ok)
        enc["tag"].append("")
    return dict(enc)


def get_extra_data_provider(model_name_or_path: str = "", processor=None):
    """-> callable(input_ids, image_grid_thw, video_grid_thw, attention_mask) -> {"position_ids": (B, 3, S)}: mRoPE ids of
    Qwen2-VL in the (bsz, 3, seqlen) layout DataProto wants (reference :330-381)."""
    tok = processor.tokenizer
    merge = processor.image_processor.merge_size
    ids = {k: tok.convert_tokens_to_ids(v) for k, v in (("image", "<|image_pad|>"), ("start", "<|vision_start|>"))}

    def extra_data_provider(input_ids, image_grid_thw=None, video_grid_thw=None, attention_mask=None):
        grids = None if image_grid_thw is None else [tuple(int(v) for v in g) for g in image_grid_thw.tolist()]
        pos, _ = hostops.get_rope_index(input_ids, grids, attention_mask, spatial_merge_size=merge, image_token_id=ids["image"],
                                        vision_start_token_id=ids["start"])
        return {"position_ids": pos.transpose(0, 1)}

    return extra_data_provider


def get_dataloader(dataset: Dict[str, List], batch_size: int, data_collator, shuffle: bool = False, seed: int = 42):
    """Batches of collated samples (the reference shuffles with torch's DataLoader; the default here is in-order so that
    runs are reproducible across rank counts)."""
    n = len(dataset["id"])
    order = np.random.default_rng(seed).permutation(n) if shuffle else np.arange(n)
    for b0 in range(0, n, batch_size):
        yield data_collator([{k: v[i] for k, v in dataset.items()} for i in order[b0:b0 + batch_size]])


def draw_visual_prompt(image, mask, visual_prompt):
    """Mask overlay (device kernel) + the prompt's box (2-px blue) and points (radius-5 discs: green positive, red
    negative) for the render1 / render2 outputs (reference :454-509)."""
    from PIL import Image, ImageDraw
    arr = torch.from_numpy(np.array(image.convert("RGB"), dtype=np.uint8, order="C")).cuda()
    try:
        m = np.asarray(mask.convert("L")) if hasattr(mask, "convert") else np.asarray(mask)
        raster.render_overlay_(arr, torch.from_numpy(np.ascontiguousarray((m > 0).astype(np.uint8))).cuda(), [])
    except Exception:  # noqa: BLE001
        pass
    out = Image.fromarray(arr.cpu().numpy())
    draw = ImageDraw.Draw(out)
    vp = visual_prompt or {}
    if "box" in vp and len(np.asarray(vp["box"]).reshape(-1)) == 4:
        b = [float(v) for v in np.asarray(vp["box"]).reshape(-1)]
        draw.rectangle([(b[0], b[1]), (b[2], b[3])], outline="blue", width=2)
    for pt, lab in zip(vp.get("point_coords", []), vp.get("point_labels", [])):
        x, y = [float(v) for v in np.asarray(pt).reshape(-1)[:2]]
        draw.ellipse([x - 5, y - 5, x + 5, y + 5], fill="green" if int(lab) == 1 else "red", outline=None)
    return out


def _obj(values) -> np.ndarray:
    a = np.empty(len(values), dtype=object)
    a[:] = list(values)
    return a


def _default_sam_provider(model_args=None, **kw):
    """seg_worker.py:540 of the reference hands ``sam2_seg_model_provider`` to the strategy.  Same here when ``seg_infer.model_args`` names a
    model (roll.models.model_providers: the MI355X SAM2); a configuration without one (unit tests of the host flow) gets the stand-in
    that turns prompts into rectangles."""
    path = model_args.get("model_name_or_path") if hasattr(model_args, "get") else getattr(model_args, "model_name_or_path", None)
    if not path:
        return socioseg_data.SyntheticSamPredictor()
    from roll.models.model_providers import sam2_seg_model_provider
    return sam2_seg_model_provider(model_args=model_args, **kw)


class SocioSegInferPipeline(BasePipeline):
    def __init__(self, pipeline_config, sam_predictor_provider=None, dataset: Optional[List[Dict]] = None, processor=None,
                 actor_worker=None):
        super().__init__(pipeline_config)
        cfg = pipeline_config
        self.rank, self.world, local = dp.init_distributed()
        # ---- workers (reference: Ray clusters actor_infer / seg_infer; here the local rank's two workers)
        self.actor_infer = actor_worker or ActorWorker(cfg.actor_infer, cfg, self.rank, self.world, local, "actor_infer")
        if actor_worker is None:
            self.actor_infer.initialize(cfg)
        self.geom = getattr(self.actor_infer.strategy, "geom", None)
        # ---- processor / tokenizer: the checkpoint's when there is one, else the offline stand-in
        if processor is None:
            path = str((cfg.actor_infer.model_args or {}).get("model_name_or_path") or cfg.get("pretrain") or "")
            if os.path.isdir(path):
                from socioreasoner_amd.textproc import load_hf_processor
                processor = load_hf_processor(path)         # AutoProcessor.from_pretrained (reference :518-521)
            else:
                from socioreasoner_amd.textproc import SyntheticProcessor
                processor = SyntheticProcessor(self.geom)
        self.processor = processor
        margs = (cfg.actor_train or {}).get("model_args") or cfg.actor_infer.model_args or {}
        self.processor.image_processor.max_pixels = int(margs.get("max_pixels") or 768 * 768)
        self.processor.image_processor.min_pixels = int(margs.get("min_pixels") or 56 * 56)
        sz = getattr(self.processor.image_processor, "size", None)
        if sz is not None and hasattr(sz, "longest_edge"):       # transformers 5 keeps the two bounds in `size`; the attributes above are what process_image reads
            sz.longest_edge, sz.shortest_edge = self.processor.image_processor.max_pixels, self.processor.image_processor.min_pixels
        self.tokenizer = self.processor.tokenizer
        self.tokenizer.padding_side = "left"
        self.actor_infer.tokenizer = self.tokenizer
        if hasattr(self.actor_infer.strategy, "tokenizer"):
            self.actor_infer.strategy.tokenizer = self.tokenizer
        self.seg_infer = SegWorker(cfg.seg_infer, cfg, self.rank, self.world, local, "seg_infer")
        self.seg_infer.initialize(cfg, model_provider=sam_predictor_provider or _default_sam_provider, tokenizer=self.tokenizer)
        # ---- data: this rank's contiguous shard (np.array_split sizes, like the reference's DP dispatch)
        if dataset is None:
            dargs = (cfg.actor_train or {}).get("data_args") or {}
            ddir = dargs.get("dataset_dir") and os.path.join(dargs.get("dataset_dir"), dargs.get("file_name") or "")
            if ddir and os.path.isdir(os.path.join(ddir, "test")):
                dataset = socioseg_data.load_socioseg_folder(ddir, "test")
            else:
                dataset = socioseg_data.synthetic_socioseg(int(os.environ.get("SOCIOSEG_NUM_SAMPLES", cfg.rollout_batch_size)))
        self.n_samples = len(dataset)
        lo, hi = dp.shard_range(self.n_samples, self.rank, self.world)
        raw = {k: [s[k] for s in dataset[lo:hi]] for k in ("id", "problem", "map_image", "sat_image", "mask_label")}
        self.dataset = encode_function(raw, self.processor)
        self.extra_data_provider = get_extra_data_provider(processor=self.processor)
        self.data_collator = DataCollatorWithPaddingForMultiSeg(
            tokenizer=self.tokenizer, processor=self.processor, extra_data_provider=self.extra_data_provider,
            max_length=int(cfg.prompt_length), image_key="image", padding="max_length", gt_object_key="gt_object", gt_bbox_key="gt_bbox")
        self.batch_size = int(os.environ.get("SOCIOSEG_BATCH", cfg.rollout_batch_size))
        self.generate_scheduler = GenerateScheduler()

    # one generation round through the scheduler -> the reference's 7 output tensors
    def _generate(self, gen_batch: DataProto, global_step: int) -> DataProto:
        gen_batch.meta_info = {"global_step": global_step, "response_callback_fn": self.generate_scheduler.report_response}
        return self.generate_scheduler.generate(data=gen_batch, actor_cluster=self.actor_infer, pipeline_config=self.pipeline_config)

    def _stage2_batch(self, batch: DataProto, bboxs_text_list: List[str]) -> DataProto:
        """Render stage 1 onto both images and build the stage-2 prompts (reference :714-825)."""
        cfg = self.pipeline_config
        padded, loose, grids = defaultdict(list), defaultdict(list), []
        ip = self.processor.image_processor
        m2 = ip.merge_size ** 2
        img_tok = "<|image_pad|>"
        for question, bboxs_text, images, mask in zip(batch.non_tensor_batch["question"], bboxs_text_list,
                                                      batch.non_tensor_batch["image"], batch.non_tensor_batch["map_mask"]):
            text = format_prompt_2(question, bboxs_text, self.processor)
            # the rendered pair stays in HBM (uint8 HWC): render kernel -> patchify kernel, no PIL / host copy in between;
            # what the processor would have contributed -- the grid and the expanded placeholders -- is computed here
            rd_image = render_image(bboxs_text, images, mask, keep_on_device=True)
            grid = torch.tensor([[1, int(t.shape[0]) // ip.patch_size, int(t.shape[1]) // ip.patch_size] for t in rd_image], dtype=torch.long)
            pieces = text.split(img_tok)
            if len(pieces) - 1 != len(rd_image):
                raise ValueError(f"text has {len(pieces) - 1} image placeholders for {len(rd_image)} images")
            full = pieces[0]
            for g_, rest in zip(grid.tolist(), pieces[1:]):
                full += img_tok * (g_[0] * g_[1] * g_[2] // m2) + rest
            ids = self.tokenizer.encode(full, add_special_tokens=False)
            padded["input_ids"].append(ids)
            padded["attention_mask"].append([1] * len(ids))
            grids.append(grid)
            loose["multi_modal_sat_inputs"].append({"image_grid_thw": grid})
            loose["multi_modal_sat_data"].append({"prompt_token_ids": self.tokenizer.encode(text, add_special_tokens=False),
                                                  "multi_modal_data": {"image": rd_image}})
        sat = self.tokenizer.pad(padded, padding="max_length", max_length=int(cfg.prompt_length), pad_to_multiple_of=None, return_tensors="pt")
        sat = {"input_ids": sat["input_ids"], "attention_mask": sat["attention_mask"]}
        extra = self.extra_data_provider(input_ids=sat["input_ids"], attention_mask=sat["attention_mask"], image_grid_thw=torch.cat(grids, dim=0))
        sat["position_ids"] = extra["position_ids"]
        sat.update({k: _obj(v) for k, v in loose.items()})
        sat["bboxs_text"] = _obj(bboxs_text_list)
        return DataProto.from_single_dict(sat)

    # ---- the host flow behind the two generations, for any set of samples: a whole rollout batch, or -- streamed mode -- the samples whose requests
    #      have just finished
    def _after_stage1(self, batch: DataProto, out: DataProto, n_ret: int, lap, dirs, writers, pending) -> DataProto:
        """stage-1 responses -> masks -> stage-1 files -> the stage-2 generation batch (reference :645-836).  `batch` is updated in place."""
        out.rename(["input_ids", "attention_mask", "position_ids", "responses", "response_mask", "prompts", "prompt_mask"],
                   ["map_input_ids", "map_attention_mask", "map_position_ids", "map_responses", "map_response_mask", "map_prompts", "map_prompt_mask"])
        out.batch.pop("prompt_id", None)
        for k, v in batch.non_tensor_batch.items():
            batch.non_tensor_batch[k] = np.repeat(v, n_ret)
        batch.batch = out.batch
        # ---- stage-1 masks: responses -> SAM prompts -> union / nearest resize on the device
        seg_batch = batch.pop(batch_keys=["map_responses", "map_prompts"], non_tensor_batch_keys=["seg_image"])
        seg_out = self.seg_infer.segment_v4_map(seg_batch)
        lap("segment_stage1")
        batch = batch.union(seg_out)
        batch.non_tensor_batch["map_mask"] = batch.non_tensor_batch.pop("mask")
        batch.non_tensor_batch["map_visual_prompt"] = batch.non_tensor_batch.pop("visual_prompt")
        batch.non_tensor_batch.pop("response_text")
        # ---- stage 2: render stage 1 onto both images, re-prompt with the boxes found
        map_response_list = self.tokenizer.batch_decode(batch.batch["map_responses"], skip_special_tokens=False)
        bboxs_text_list = [parse_points_text_from_content(r) for r in map_response_list]
        # stage 1's files exist from here on: they are rendered (device overlay, this thread) and handed to the writer pool NOW, so that their PNG
        # encoding runs under stage 2's generation instead of after it (same files, same contents as the reference's write at the end of the batch)
        from PIL import Image
        nt1 = batch.non_tensor_batch
        for i in range(len(batch)):
            vp1 = nt1["map_visual_prompt"][i][0] if len(nt1["map_visual_prompt"][i]) else {}
            r1 = draw_visual_prompt(nt1["seg_image"][i], nt1["map_mask"][i], vp1)

            def write1(sid=nt1["id"][i], m1=nt1["map_mask"][i], r1=r1, t1=map_response_list[i]):
                Image.fromarray(m1.astype(np.uint8) * 255).save(os.path.join(dirs["stage1"], f"{sid}.png"))
                r1.save(os.path.join(dirs["render1"], f"{sid}.png"))
                with open(os.path.join(dirs["stage1"], f"{sid}.txt"), "w") as f:
                    f.write(t1)
            pending.append(writers.submit(write1))
        batch = batch.union(self._stage2_batch(batch, bboxs_text_list))
        gen_batch = batch.pop(batch_keys=["input_ids", "attention_mask", "position_ids"], non_tensor_batch_keys=["multi_modal_sat_data"])
        gen_batch.non_tensor_batch["multi_modal_data"] = gen_batch.non_tensor_batch.pop("multi_modal_sat_data")
        lap("stage2_prompts")
        return gen_batch

    def _after_stage2(self, batch: DataProto, out: DataProto, lap, dirs, writers, pending) -> List[float]:
        """stage-2 responses -> masks -> IoU against the ground truth, stage-2 files (reference :853-905)"""
        from PIL import Image
        out.batch.pop("prompt_id", None)
        batch = batch.union(out)
        seg_batch = batch.pop(batch_keys=["responses", "prompts"], non_tensor_batch_keys=["seg_image"])
        seg_out = self.seg_infer.segment_v4_sat(seg_batch)
        lap("segment_stage2")
        seg_out.meta_info.pop("metrics", None)
        batch = batch.union(seg_out)
        batch.non_tensor_batch["sat_mask"] = batch.non_tensor_batch.pop("mask")
        batch.non_tensor_batch["sat_visual_prompt"] = batch.non_tensor_batch.pop("visual_prompt")
        # ---- score and write
        sat_response_list = self.tokenizer.batch_decode(batch.batch["responses"], skip_special_tokens=False)
        nt = batch.non_tensor_batch
        giou_list = []
        for i in range(len(batch)):
            gt_mask = np.array(nt["gt_mask"][i].convert("L"))
            giou_list.append(compute_giou(nt["sat_mask"][i], gt_mask))
            vp2 = nt["sat_visual_prompt"][i][0] if len(nt["sat_visual_prompt"][i]) else {}
            sid = nt["id"][i]
            # the overlays run on the device from THIS thread (the HIP current device is per thread); the pool only encodes and writes
            r2 = draw_visual_prompt(nt["seg_image"][i], nt["sat_mask"][i], vp2)

            def write(sid=sid, m2=nt["sat_mask"][i], r2=r2, t2=sat_response_list[i]):
                Image.fromarray(m2.astype(np.uint8) * 255).save(os.path.join(dirs["stage2"], f"{sid}.png"))
                r2.save(os.path.join(dirs["render2"], f"{sid}.png"))
                with open(os.path.join(dirs["stage2"], f"{sid}.txt"), "w") as f:
                    f.write(t2)
            pending.append(writers.submit(write))
        lap("score_and_write")
        return giou_list

    def _streamed(self, n_ret: int) -> bool:
        """Streamed mode (round 6): the two stages of a rollout batch share ONE open request stream on the engine -- a sample's stage-2 prompt is
        added as soon as ITS stage-1 answer has been segmented and rendered, while other samples are still in stage 1 -- so the engine's batch rows
        never drain between the stages and the host flow (SAM2 decoding, rendering, prompt building, scoring, file encoding) runs under generation
        instead of between the two generate calls.  Per sample the operations, their order and the files are those of the batch flow.
        Needs a strategy whose request loop can stay open (`request_stream`) and a rollout batch of at least two waves of the engine's rows (SOCIOSEG_STREAM=1 forces
        it, SOCIOSEG_STREAM=0 restores the reference's batch-by-batch order).
        Not with n > 1 sequences per prompt, nor with request-level dispatch across ranks (generate_opt_level >= 1 and more than one rank: that exchange
        is collective per generate call)."""
        flag = os.environ.get("SOCIOSEG_STREAM", "auto")
        if flag == "0" or n_ret != 1:
            return False
        if int(self.pipeline_config.get("generate_opt_level") or 0) >= 1 and self.world > 1:
            return False
        st = getattr(self.actor_infer, "strategy", None)
        if not (bool(getattr(st, "request_stream", False)) and hasattr(self.actor_infer, "start_server")):
            return False
        # auto (SOCIOSEG_STREAM unset): only where there is something to overlap -- a rollout batch of at least two waves of the engine's batch rows.  With one wave
        # per stage (64 samples in rollout batches of 32 through 32 rows) every sample of a stage finishes in the same decode step, the stage-2 prompts arrive in
        # bursts the scheduler cuts into part-filled admissions, and the streamed order measured 4.3-4.9 s against 4.6-4.7 s (profiles/r06_pipeline64_ab.txt);
        # SOCIOSEG_STREAM=1 forces it
        rows = getattr(st, "max_batch", None)
        if flag != "1" and rows and self.batch_size < 2 * int(rows):
            return False
        return True

    @staticmethod
    def _rows(data: DataProto, idx: List[int]) -> DataProto:
        ix = np.asarray(idx, dtype=np.int64)
        return DataProto(batch={k: v[torch.from_numpy(ix)] for k, v in (data.batch or {}).items()},
                         non_tensor_batch={k: v[ix] for k, v in data.non_tensor_batch.items()}, meta_info=dict(data.meta_info))

    def _run_streamed(self, batches, make_parts, lap, dirs, writers, pending, on_batch_start=None) -> List[List[float]]:
        """The streamed flow over ALL rollout batches on one open request stream.
        batches: [(first sample, one past the last)] in dataset order; make_parts(k) -> iterator of (batch, stage-1 generation batch) pieces of batch k in
        sample order (the first piece's prompts are on the engine while the later pieces are still being collated).  Request ids: batch k owns
        base_k .. base_k + 2 B_k - 1 -- first its B_k stage-1 prompts, then the stage-2 prompt of sample id - base_k - B_k.  Batch k + 1 is STARTED (collated, its
        stage-1 prompts added behind whatever is queued) as soon as every stage-1 answer of batch k has been turned into a stage-2 prompt: the engine goes from
        batch k's stage 2 into batch k + 1's stage 1 without draining, and the next batch is collated under generation.  Per-batch scores come back in batch order."""
        from roll.distributed.scheduler.generate_scheduler import assemble_responses
        if not batches:
            return []
        host_stream = None
        if torch.cuda.is_available():
            # this thread's device work (SAM2 decoder, raster kernels, render) runs next to the engine's: on a stream of its own, not on the null
            # stream, which would serialise with the scheduler's CU-masked (blocking) streams
            # (a high-priority stream was measured too and changes nothing: a segment call next to the engine waits for SAM2 ENCODER passes -- its own
            # or the prefetch thread's -- not for launch slots)
            if getattr(self, "_host_stream", None) is None:
                self._host_stream = torch.cuda.Stream()
            host_stream = self._host_stream
        stream = self.generate_scheduler.open_stream(self.actor_infer, self.pipeline_config)
        live: Dict[int, dict] = {}          # batch index -> its state while any of its samples is in flight
        bases = []
        for lo, hi in batches:
            bases.append(2 * lo)            # (ids of batch k: 2 lo_k .. 2 hi_k - 1)
        results: Dict[int, List[float]] = {}
        started = 0

        def start(k):
            lo, hi = batches[k]
            B = hi - lo
            if on_batch_start is not None:
                on_batch_start(k)
            pieces, n0 = [], 0
            for part, gen_part in make_parts(k):
                lap("collate")
                stream.add(list(range(bases[k] + n0, bases[k] + n0 + len(part))), gen_part)
                lap("generate_stage1")
                pieces.append((part, gen_part))
                n0 += len(part)
            assert n0 == B, (n0, B)
            live[k] = {"B": B, "s1_left": B,
                       "batch": pieces[0][0] if len(pieces) == 1 else DataProto.concat([p_ for p_, _ in pieces]),
                       "gen_batch": pieces[0][1] if len(pieces) == 1 else DataProto.concat([g_ for _, g_ in pieces]),
                       "state": {}, "gen2": {}, "giou": {}}       # state / gen2: sample -> its one-row batch after stage 1 / its stage-2 prompt row
            lap("collate")

        try:
            start(0)
            started = 1
            while len(results) < len(batches):
                got = stream.collect()
                lap("generate_wait")
                for k in sorted(live):
                    b, B, base = live[k], live[k]["B"], bases[k]
                    mine = [(rid - base, toks) for rid, toks in got if base <= rid < base + 2 * B]
                    s1 = sorted((i, t) for i, t in mine if i < B)
                    s2 = sorted((i - B, t) for i, t in mine if i >= B)
                    with (torch.cuda.stream(host_stream) if host_stream is not None else contextlib.nullcontext()):
                        if s1:
                            idx = [i for i, _ in s1]
                            sub, gsub = self._rows(b["batch"], idx), self._rows(b["gen_batch"], idx)
                            out = assemble_responses(gsub, [t for _, t in s1], self.actor_infer, self.pipeline_config)
                            gen2 = self._after_stage1(sub, out, 1, lap, dirs, writers, pending)
                            if host_stream is not None:
                                host_stream.synchronize()                # the rendered pairs are read by the engine's streams (another thread)
                            stream.add([base + B + i for i in idx], gen2)
                            for j, i in enumerate(idx):
                                b["state"][i], b["gen2"][i] = self._rows(sub, [j]), self._rows(gen2, [j])
                            b["s1_left"] -= len(idx)
                            lap("stage2_prompts")
                        if s2:
                            idx = [i for i, _ in s2]
                            sub = DataProto.concat([b["state"].pop(i) for i in idx])
                            gsub = DataProto.concat([b["gen2"].pop(i) for i in idx])
                            out = assemble_responses(gsub, [t for _, t in s2], self.actor_infer, self.pipeline_config)
                            for i, g_ in zip(idx, self._after_stage2(sub, out, lap, dirs, writers, pending)):
                                b["giou"][i] = g_
                for k in sorted(live):
                    if len(live[k]["giou"]) == live[k]["B"]:
                        results[k] = [live[k]["giou"][i] for i in range(live[k]["B"])]
                        del live[k]
                if started < len(batches) and (started - 1 not in live or live[started - 1]["s1_left"] == 0):
                    start(started)
                    started += 1
        finally:
            stream.close()
        return [results[k] for k in range(len(batches))]

    @torch.no_grad()
    def run(self):
        cfg = self.pipeline_config
        res_dir = os.path.join(cfg.output_dir, "result")
        dirs = {k: os.path.join(res_dir, k) for k in ("stage1", "stage2", "render1", "render2")}
        for d in dirs.values():
            os.makedirs(d, exist_ok=True)
        n_ret = int((cfg.actor_infer.generating_args or {}).get("num_return_sequences", 1) or 1)
        all_giou: List[float] = []
        global_step = 0
        import time as _time
        from concurrent.futures import ThreadPoolExecutor
        self.streamed = self._streamed(n_ret)
        self.timing = {k: 0.0 for k in ("collate", "generate_stage1", "segment_stage1", "stage2_prompts", "generate_stage2", "segment_stage2", "score_and_write")}
        if self.streamed:       # the generations run under the other phases: what is left of them on this thread is the wait for the next finished request
            self.timing["generate_wait"] = 0.0
        clock = [_time.perf_counter()]

        def lap(name):          # wall time since the previous lap (the reference times generate / seg with _Timer: ...infer.py:625-704)
            now = _time.perf_counter()
            self.timing[name] += now - clock[0]
            clock[0] = now
        # PNG encoding of four 768 x 768 images per sample was the largest single term of a batch on the driver (zlib; PIL releases the GIL):
        # the files are written by a small pool while the next batch generates; run() returns after the last one is on disk
        writers = ThreadPoolExecutor(max_workers=int(os.environ.get("SOCIOSEG_WRITERS", 8)))
        pending = []
        prefetch = os.environ.get("SOCIOSEG_SAM_PREFETCH", "1") != "0" and hasattr(self.seg_infer, "prefetch_images")
        import sys
        switch = sys.getswitchinterval()
        n_all = len(self.dataset["id"])
        spans = [(b0, min(b0 + self.batch_size, n_all)) for b0 in range(0, n_all, self.batch_size)]       # (in dataset order: batch k = samples k * batch_size ...)

        def stage1_batches(batch_dict, step):
            batch = DataProto.from_single_dict(batch_dict)
            batch.meta_info = {"global_step": step}
            # ---- stage 1: generate on the (map, satellite) pair
            gen_batch = batch.pop(batch_keys=["map_input_ids", "map_attention_mask", "map_position_ids"], non_tensor_batch_keys=["multi_modal_map_data"])
            gen_batch.rename(["map_input_ids", "map_attention_mask", "map_position_ids"], ["input_ids", "attention_mask", "position_ids"])
            gen_batch.non_tensor_batch["multi_modal_data"] = gen_batch.non_tensor_batch.pop("multi_modal_map_data")
            return batch, gen_batch

        def start_prefetch(k):
            if prefetch:
                # SAM2's set_image needs only the pixels (seg_strategy.py:47-58): its encoder starts now, under the collation of the batch and the
                # LM's stage-1 generation
                self.seg_infer.prefetch_images(list(self.dataset["seg_image"][spans[k][0]:spans[k][1]]))

        if self.streamed:
            # three Python threads share the interpreter in streamed mode (this one, the request loop, SAM2's prefetch): the request loop needs it for
            # ~1.5 ms between two chunks of decode steps, and every 5 ms (the default switch interval) it waits for it is a chunk the GPU starts late
            sys.setswitchinterval(0.0005)
            piece = max(1, int(os.environ.get("SOCIOSEG_COLLATE_CHUNK", 32)))
            model = getattr(getattr(self.seg_infer, "strategy", None), "model", None)
            if hasattr(model, "cache_images"):      # two rollout batches are in flight at a batch boundary: batch k's stage 2 still needs its embeddings while batch k + 1's arrive
                model.cache_images = max(int(model.cache_images), 2 * self.batch_size + 32)

            def make_parts(k):
                # the rollout batch is collated in pieces (every row is padded to prompt_length and indexed on its own, so the pieces ARE the rows of the
                # whole batch's collation): the first piece's prompts are being prefilled while the rest is collated
                lo, hi = spans[k]
                return (stage1_batches(self.data_collator([{key: v[i] for key, v in self.dataset.items()} for i in range(c0, min(c0 + piece, hi))]), k)
                        for c0 in range(lo, hi, piece))

            def on_batch_start(k):
                self.model_update(k)
                start_prefetch(k)
            try:
                for giou_list in self._run_streamed(spans, make_parts, lap, dirs, writers, pending, on_batch_start):
                    print(f"giou_acc: {np.mean(giou_list)}")
                    all_giou.extend(giou_list)
                    global_step += 1
            finally:
                sys.setswitchinterval(switch)
            lap("score_and_write")
        else:
            loader = get_dataloader(self.dataset, self.batch_size, self.data_collator)
            for k in range(len(spans)):
                start_prefetch(k)
                batch_dict = next(loader)
                lap("collate")
                self.model_update(global_step)
                batch, gen_batch = stage1_batches(batch_dict, global_step)
                out = self._generate(gen_batch, global_step)
                lap("generate_stage1")
                gen_batch = self._after_stage1(batch, out, n_ret, lap, dirs, writers, pending)
                ga = cfg.actor_infer.generating_args or {}
                keep = ga.get("num_return_sequences", 1)
                ga["num_return_sequences"] = 1                      # stage 2 never fans out (reference :838-852)
                lap("stage2_prompts")
                out = self._generate(gen_batch, global_step)
                lap("generate_stage2")
                ga["num_return_sequences"] = keep
                giou_list = self._after_stage2(batch, out, lap, dirs, writers, pending)
                print(f"giou_acc: {np.mean(giou_list)}")
                all_giou.extend(giou_list)
                global_step += 1
                lap("score_and_write")
        if prefetch and hasattr(self.seg_infer, "wait_prefetch"):
            self.seg_infer.wait_prefetch()          # (segment calls no longer join the prefetch thread: nothing of it may outlive run() -- the caller may close the engines next)
        # request-level dispatch across ranks is collective: a rank whose shard had fewer batches than the largest shard joins the
        # other ranks' remaining rounds (two generate calls per batch) with no requests of its own
        most = max(-(-s_ // self.batch_size) for s_ in dp.split_sizes(self.n_samples, self.world))
        for _ in range(2 * (most - global_step)):
            self.generate_scheduler.join_idle_round(self.actor_infer, self.pipeline_config)
        for fut in pending:
            fut.result()                # (re-raises a failed write)
        writers.shutdown()
        lap("score_and_write")
        local = torch.tensor(all_giou, dtype=torch.float64).reshape(-1, 1)
        if self.world > 1:
            # samples are array_split over the ranks and every sample contributes n_ret rows
            local = dp.all_gather_rows(local.cuda() if torch.cuda.is_available() else local, self.n_samples * n_ret,
                                       sizes=[s_ * n_ret for s_ in dp.split_sizes(self.n_samples, self.world)]).cpu()
        giou_acc = float(local.mean()) if local.numel() else 0.0
        if self.rank == 0:
            print(f"giou_acc: {giou_acc}")
            with open(os.path.join(res_dir, "iou_acc.txt"), "w") as f:
                f.write(f"giou_acc: {giou_acc}")
        return giou_acc
