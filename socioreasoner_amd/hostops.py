"""Host-side integer logic of the hot path, product side (the reference keeps all of this in Python too).

Mirrors, with the same names and argument meaning (paths relative to /root/reference):
  smart_resize                      hf:models/qwen2_vl/image_processing_pil_qwen2_vl.py:57-83 as bound at
                                    roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:126-144, 518-521
  get_rope_index                    mcore_adapter/src/mcore_adapter/models/qwen2_5_vl/modeling_qwen2_5_vl.py:319-441
  gather_unpadded_input_ids         roll/distributed/strategy/vllm_strategy.py:274-276
  gather_outputs_to_pad_tensor      roll/distributed/strategy/vllm_strategy.py:279-286
  concatenate_input_and_output      roll/utils/functionals.py:364-373
  pad_to_length / get_pad_mask      roll/utils/functionals.py:351-361, 301-313
  postprocess_generate              roll/utils/functionals.py:768-872
  parse_points_text_from_content    roll/pipeline/multi_utils.py:4-15
  parse_visual_prompt_from_json_s2  roll/pipeline/rlvr/seg_worker.py:199-259
  compute_giou                      roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:45-58
"""
from __future__ import annotations

import json
import math
import re
from typing import Any, Dict, List

import numpy as np
import torch


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 56 * 56, max_pixels: int = 768 * 768):
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar, w_bar = round(height / factor) * factor, round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar, w_bar = math.ceil(height * beta / factor) * factor, math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def get_rope_index(input_ids: torch.Tensor, image_grid_thw=None, attention_mask=None, *, spatial_merge_size=2,
                   image_token_id=151655, vision_start_token_id=151652):
    """mRoPE position ids.  input_ids [B,S] -> (position_ids [3,B,S] int64, mrope_position_deltas [B,1]).
    Images only (second_per_grid_t = 0), which is all the infer pipeline feeds."""
    input_ids = torch.as_tensor(input_ids)
    B, S = input_ids.shape
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    attention_mask = torch.as_tensor(attention_mask)
    if image_grid_thw is None or len(image_grid_thw) == 0:
        pos = attention_mask.long().cumsum(-1) - 1
        pos.masked_fill_(attention_mask == 0, 1)
        pos = pos.unsqueeze(0).expand(3, -1, -1).clone()
        deltas = pos.max(0)[0].max(-1, keepdim=True)[0] + 1 - S
        return pos, deltas
    grids = [tuple(int(x) for x in g) for g in torch.as_tensor(image_grid_thw).tolist()]
    position_ids = torch.ones(3, B, S, dtype=torch.long)
    deltas, gi = [], 0
    for b in range(B):
        keep = attention_mask[b] == 1
        ids = input_ids[b][keep]
        n = ids.numel()
        out = torch.empty(3, n, dtype=torch.long)
        is_img = ids == image_token_id
        # every image = one maximal run of image tokens preceded by <vision_start>
        run_starts = torch.nonzero(is_img & ~torch.cat([torch.zeros(1, dtype=torch.bool), is_img[:-1]])).flatten().tolist()
        run_starts = [s for s in run_starts if s > 0 and int(ids[s - 1]) == vision_start_token_id]
        cursor, nxt = 0, 0                     # cursor: index into ids, nxt: next free position value
        for s in run_starts:
            t, h, w = grids[gi]
            gi += 1
            lh, lw = h // spatial_merge_size, w // spatial_merge_size
            ntxt = s - cursor
            out[:, cursor:s] = torch.arange(ntxt) + nxt
            base = nxt + ntxt
            k = t * lh * lw
            hh = torch.arange(lh).repeat_interleave(lw).repeat(t)
            ww = torch.arange(lw).repeat(lh * t)
            out[0, s:s + k] = base
            out[1, s:s + k] = hh + base
            out[2, s:s + k] = ww + base
            nxt = int(out[:, cursor:s + k].max()) + 1 if s + k > cursor else nxt
            cursor = s + k
        if cursor < n:
            out[:, cursor:] = torch.arange(n - cursor) + nxt
        position_ids[:, b, keep] = out
        deltas.append(int(out.max()) + 1 - S)
    return position_ids, torch.tensor(deltas).unsqueeze(1)


def rope_index_1d(ids, grids=None, *, spatial_merge_size=2, image_token_id=151655, vision_start_token_id=151652):
    """`get_rope_index` for ONE unpadded sequence, in numpy: ids [S] -> position ids [3, S] int64.  Same rule (text tokens count up from the largest
    position used so far + 1; image k's tokens take (base, base + row, base + column) of its merged grid) -- the per-request form the strategy
    needs: the torch version above costs ~5 ms per call on a many-core host (dozens of tiny tensor ops), this one ~0.1 ms.  Pinned to it by
    tests/test_host_round6.py on random prompt structures."""
    import numpy as np
    ids = np.asarray(ids, dtype=np.int64)
    n = ids.shape[0]
    out = np.empty((3, n), dtype=np.int64)
    if not grids:
        out[:] = np.arange(n, dtype=np.int64)
        return out
    is_img = ids == image_token_id
    prev = np.concatenate([[False], is_img[:-1]])
    starts = [int(s_) for s_ in np.nonzero(is_img & ~prev)[0] if s_ > 0 and int(ids[s_ - 1]) == vision_start_token_id]
    cursor, nxt, gi = 0, 0, 0
    for s_ in starts:
        t, h, w = (int(v) for v in grids[gi])
        gi += 1
        lh, lw = h // spatial_merge_size, w // spatial_merge_size
        ntxt = s_ - cursor
        out[:, cursor:s_] = np.arange(ntxt, dtype=np.int64) + nxt
        base = nxt + ntxt
        k = t * lh * lw
        out[0, s_:s_ + k] = base
        out[1, s_:s_ + k] = np.tile(np.repeat(np.arange(lh, dtype=np.int64), lw), t) + base
        out[2, s_:s_ + k] = np.tile(np.arange(lw, dtype=np.int64), lh * t) + base
        if s_ + k > cursor:
            nxt = int(out[:, cursor:s_ + k].max()) + 1
        cursor = s_ + k
    if cursor < n:
        out[:, cursor:] = np.arange(n - cursor, dtype=np.int64) + nxt
    return out


def gather_unpadded_input_ids(input_ids: torch.Tensor, attention_mask: torch.Tensor):
    return [ids[mask.bool()].tolist() for ids, mask in zip(input_ids, attention_mask)]


def gather_outputs_to_pad_tensor(token_ids_list: List[List[int]], pad_token_id: int, device="cpu") -> torch.Tensor:
    L_ = max((len(t) for t in token_ids_list), default=0)
    out = torch.full((len(token_ids_list), L_), pad_token_id, dtype=torch.long, device=device)
    for i, t in enumerate(token_ids_list):
        out[i, : len(t)] = torch.as_tensor(t, dtype=torch.long, device=device)
    return out


def concatenate_input_and_output(input_ids: torch.Tensor, output_ids: torch.Tensor, num_return_sequences: int):
    b, s = input_ids.shape
    rep = input_ids.unsqueeze(1).repeat(1, num_return_sequences, 1).view(b * num_return_sequences, s)
    return torch.cat((rep, output_ids), dim=1)


def pad_to_length(tensor: torch.Tensor, length: int, pad_value, dim: int = -1):
    if tensor.size(dim) >= length:
        idx = [slice(None)] * tensor.ndim
        idx[dim] = slice(0, length)
        return tensor[tuple(idx)]
    shape = list(tensor.shape)
    shape[dim] = length - tensor.size(dim)
    return torch.cat([tensor, torch.full(shape, pad_value, dtype=tensor.dtype, device=tensor.device)], dim=dim)


def get_pad_mask(response_id: torch.Tensor, pad_token: int = 0, dtype=torch.int64):
    pad_mask = response_id.not_equal(pad_token).to(dtype)
    assert not (pad_mask[:, 0] == 0).logical_and(pad_mask.sum(-1) != 0).any(), \
        f"response_id is not valid: {response_id}, pad_token is {pad_token}"
    return pad_mask


def postprocess_generate(prompts: Dict[str, torch.Tensor], output: torch.Tensor, num_return_sequences: int,
                         sequence_length: int, eos_token_id: int, pad_token_id: int, fill_eos_token: bool = False):
    """Same tensors, names and left->right-pad conversion as the reference; ``prompts`` is a dict with
    input_ids / attention_mask / position_ids (the DataProto.batch of the reference)."""
    output = output.clone()
    if fill_eos_token:
        last = output.size(1) - 1
        need = output[:, last] != pad_token_id
        output[need, last] = eos_token_id
    input_ids, attention_mask = prompts["input_ids"], prompts["attention_mask"]
    obs, P = output.size(0), input_ids.size(1)
    output = pad_to_length(output, sequence_length, pad_token_id)
    prompt, response = output[:, :P].clone(), output[:, P:].clone()
    attention_mask = attention_mask.unsqueeze(1).repeat(1, num_return_sequences, 1).view(obs, P)
    response_mask = get_pad_mask(response, pad_token_id, attention_mask.dtype)
    attention_mask = torch.cat((attention_mask, response_mask), dim=-1)
    position_ids = prompts["position_ids"]
    mrope = position_ids.dim() == 3
    if mrope:
        position_ids = position_ids.unsqueeze(1).repeat(1, num_return_sequences, 1, 1).view(obs, *position_ids.shape[-2:])
        delta = torch.arange(1, sequence_length - P + 1, device=position_ids.device).view(1, 1, -1).expand(obs, 3, -1)
        out_pos = torch.cat([position_ids, position_ids[..., -1:] + delta], dim=-1)
    assert attention_mask.any(dim=1).all(), "has all 0 attention_mask"
    first_one = attention_mask.float().argmax(dim=1)
    new_response_mask = torch.zeros_like(attention_mask)
    for i in range(obs):
        shift = int(first_one[i])
        if shift > 0:
            output[i, :-shift] = output[i, shift:].clone()
        valid = int(attention_mask[i].sum())
        rl = int(response_mask[i].sum())
        attention_mask[i][:valid] = 1
        attention_mask[i][valid:] = 0
        new_response_mask[i][valid - rl: valid] = 1
        if mrope and shift > 0:
            out_pos[i, ..., :-shift] = out_pos[i, ..., shift:].clone()
            if P > rl:
                output[i, -shift:] = pad_token_id
    prompt_mask = (attention_mask == 1) & (new_response_mask == 0)
    position_ids = out_pos if mrope else torch.clip(torch.cumsum(attention_mask, dim=-1) - 1, min=0, max=None)
    return {"prompts": prompt, "responses": response, "input_ids": output, "attention_mask": attention_mask,
            "position_ids": position_ids, "prompt_mask": prompt_mask, "response_mask": new_response_mask}


_ANSWER_RE = re.compile(r"<answer>(.*?)</answer>", re.DOTALL)


def parse_points_text_from_content(content: str) -> str:
    m = _ANSWER_RE.search(content)
    return m.group(1).strip() if m else ""


def parse_visual_prompt_from_json_s2(content: str) -> List[Dict[str, Any]]:
    parsed: List[Dict[str, Any]] = []
    m = _ANSWER_RE.search(content)
    if not m:
        return parsed
    try:
        data = json.loads(m.group(1).strip())
    except json.JSONDecodeError:
        return parsed
    if not isinstance(data, list):
        return []
    for obj in data:
        try:
            if not isinstance(obj, dict):
                continue
            box = obj.get("bbox_2d", [])
            points = [[p[0], p[1]] for p in obj.get("points", [])]
            labels = np.ones(len(points), dtype=int).tolist()
            if isinstance(box, list) and len(box) == 4:
                parsed.append({"box": box, "points": points, "labels": labels})
        except Exception:
            continue
    return parsed


def compute_giou(pred_mask: np.ndarray, gt_mask: np.ndarray) -> float:
    """CPU form kept for callers that hold numpy masks; the device path is raster.iou_counts."""
    p, g = pred_mask > 0, gt_mask > 0
    union = np.logical_or(p, g).sum()
    if union == 0:
        return 1.0
    return np.logical_and(p, g).sum() / union


def get_dist_info_from_comm_plan(comm_plan: Dict[Any, Dict[str, Any]], rank_in_cluster: int, rank_in_worker: int):
    """Which broadcast tree of a model-update comm plan this (engine worker, device) belongs to and at which group rank
    (reference roll/utils/functionals.py:875-882): group rank 0 is the sending trainer rank, rank i >= 1 is the i-th entry of
    ``tgt_devices`` = {"rank": worker rank in the target cluster, "device": {"rank": device index inside that worker, ...}}.
    -> (group rank, that tree's comm_plan_args) or (None, None)."""
    for args in comm_plan.values():
        for pos, tgt in enumerate(args["tgt_devices"], start=1):
            if tgt["rank"] == rank_in_cluster and tgt["device"]["rank"] == rank_in_worker:
                return pos, args
    return None, None
