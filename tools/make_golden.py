#!/usr/bin/env python3
"""Generate tests/golden/* -- runs ONLY in the build container.

Sources of truth executed here (nothing from them is copied into the repo, only inputs/outputs are stored):
  * the un-vendored dependency that holds the model arithmetic: ``transformers`` 5.15.0
    (Qwen2_5_VLForConditionalGeneration, bf16, eager attention) and its PIL image processor;
  * the reference's own pure functions, loaded at run time from /root/reference by AST extraction
    (get_rope_index, postprocess_generate & friends, compute_giou, the two parsers).  The two that
    touch TensorDict/DataProto get trivial stand-ins *in this process only*;
  * PIL 12.2 for the render (alpha_composite / ImageDraw.rectangle) rule.
cv2 and SAM2 are not installed: INTER_NEAREST and the SAM2 forward stay unpinned.

Usage: python tools/make_golden.py            (writes tests/golden/*.npz|json)
"""
from __future__ import annotations

import ast
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import model_ref as M  # noqa: E402
from oracle import weights as WG  # noqa: E402


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def extract(path, names, ns):
    """exec the named top-level functions (or methods of any class) of a reference file into ``ns``."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names and node.name not in found:
            found[node.name] = ast.get_source_segment(src, node)
    for n in names:
        code = found[n]
        lines = code.split("\n")
        indent = len(lines[0]) - len(lines[0].lstrip())
        code = "\n".join(l[indent:] if len(l) >= indent else l for l in lines)
        exec(compile(code, path + ":" + n, "exec"), ns)
    return ns


# ----------------------------------------------------------------------------- HF model helpers
def hf_config(cfg):
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLConfig
    v, t = cfg.vision, cfg.text
    c = Qwen2_5_VLConfig(
        vision_config=dict(depth=v.depth, hidden_size=v.hidden_size, num_heads=v.num_heads,
                           intermediate_size=v.intermediate_size, patch_size=v.patch_size,
                           temporal_patch_size=v.temporal_patch_size, spatial_merge_size=v.spatial_merge_size,
                           window_size=v.window_size, fullatt_block_indexes=list(v.fullatt_block_indexes),
                           out_hidden_size=v.out_hidden_size, hidden_act="silu"),
        text_config=dict(num_hidden_layers=t.num_hidden_layers, hidden_size=t.hidden_size,
                         num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                         intermediate_size=t.intermediate_size, vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps,
                         rope_parameters={"rope_type": "default", "rope_theta": t.rope_theta,
                                          "mrope_section": list(t.mrope_section)},
                         max_position_embeddings=32768, tie_word_embeddings=True,
                         bos_token_id=None, eos_token_id=None),
        image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id,
        vision_start_token_id=cfg.vision_start_token_id, vision_end_token_id=cfg.vision_end_token_id,
        tie_word_embeddings=True)
    c._attn_implementation = "eager"
    return c


def hf_model(cfg, W):
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VLForConditionalGeneration
    model = Qwen2_5_VLForConditionalGeneration(hf_config(cfg)).to(torch.bfloat16).eval()
    sd = model.state_dict()
    for name in sd:
        ref = "model.embed_tokens.weight" if name == "lm_head.weight" else \
            name.replace("model.visual.", "visual.").replace("model.language_model.", "model.")
        sd[name].copy_(W[ref].reshape(sd[name].shape).to(torch.bfloat16))
    # ``.to(bfloat16)`` also rounded the non-persistent rotary ``inv_freq`` buffers; ``from_pretrained(dtype=bf16)``
    # (the reference's load path) keeps them float32 because they are created with an explicit float dtype
    # (hf:130-131, 521-522).  Restore that state.
    for mod in model.modules():
        if hasattr(mod, "inv_freq"):
            if hasattr(mod, "original_inv_freq"):
                inv, _ = mod.compute_default_rope_parameters(mod.config)
                mod.original_inv_freq = inv.clone()
            else:
                inv = 1.0 / (mod.theta ** (torch.arange(0, mod.dim, 2, dtype=torch.float) / mod.dim))
            mod.inv_freq = inv.float()
    return model


def synth_input(name, shape, scale=1.0):
    """Inputs come from the same counter hash as the weights (std 0.02) scaled up -> regenerable anywhere."""
    return torch.from_numpy(WG.synth_f32("input." + name, shape, seed=7)) * (scale / 0.02)


def build_prompt(cfg, grids, n_pre, n_mid, n_post, seed):
    """ids = pre text, then per image <vs><pad>*T<ve> + mid text, then post text."""
    rng = np.random.default_rng(seed)
    hi = min(cfg.image_token_id, cfg.text.vocab_size) - 8
    parts = [rng.integers(0, hi, n_pre)]
    for (t, h, w) in grids:
        T = t * h * w // 4
        parts += [[cfg.vision_start_token_id], [cfg.image_token_id] * T, [cfg.vision_end_token_id],
                  rng.integers(0, hi, n_mid)]
    parts.append(rng.integers(0, hi, n_post))
    return np.concatenate([np.asarray(p, dtype=np.int64) for p in parts])


# ----------------------------------------------------------------------------- fixtures
def gen_hf_tiny(ref_rope):
    cfg = M.config_tiny()
    W = WG.LazyWeights(cfg, seed=0)
    model = hf_model(cfg, W)
    grids = [(1, 12, 8), (1, 8, 8)]
    N = sum(t * h * w for t, h, w in grids)
    pix = synth_input("tiny.pix", (N, 1176), scale=1.0)
    ids = build_prompt(cfg, grids, 7, 3, 9, seed=11)
    S = len(ids)
    pos3, _ = ref_rope(cfg, torch.tensor(ids)[None], torch.tensor(grids), torch.ones(1, S, dtype=torch.long))
    with torch.no_grad():
        vis = model.model.visual(pix.to(torch.bfloat16), torch.tensor(grids))
        out = model(input_ids=torch.tensor(ids)[None], attention_mask=torch.ones(1, S, dtype=torch.long),
                    position_ids=pos3, pixel_values=pix.to(torch.bfloat16), image_grid_thw=torch.tensor(grids),
                    use_cache=False)
        # single-op stages on identical inputs (ViT block 0, window attention; LM layer 0)
        vm = model.model.visual
        blk = vm.blocks[0]
        x_in = synth_input("tiny.vit_x", (N, cfg.vision.hidden_size), scale=1.0).to(torch.bfloat16)
        widx, cu_win = M.vision_window_index(grids, 2, 112, 14)
        cos, sin = M.vit_rotary_tables(cfg.vision, grids, widx)
        st = {
            "vit_x": bf16_bits(x_in),
            "vit_norm1": bf16_bits(blk.norm1(x_in)),
            "vit_qkv": bf16_bits(blk.attn.qkv(x_in)),
            "vit_attn_win": bf16_bits(blk.attn(x_in, cu_seqlens=cu_win.to(torch.int32), position_embeddings=(cos, sin))),
            "vit_attn_full": bf16_bits(blk.attn(x_in, cu_seqlens=M.vision_full_seqlens(grids).to(torch.int32),
                                                position_embeddings=(cos, sin))),
            "vit_mlp": bf16_bits(blk.mlp(x_in)),
            "vit_block_win": bf16_bits(blk(x_in, cu_seqlens=cu_win.to(torch.int32), position_embeddings=(cos, sin))),
            "vit_merger": bf16_bits(vm.merger(x_in)),
            "vit_patch_embed": bf16_bits(vm.patch_embed(pix.to(torch.bfloat16))),
        }
        lm = model.model.language_model
        lx = synth_input("tiny.lm_x", (1, S, cfg.text.hidden_size), scale=1.0).to(torch.bfloat16)
        pe = lm.rotary_emb(lx, pos3)
        from transformers.masking_utils import create_causal_mask
        mask = create_causal_mask(config=lm.config, inputs_embeds=lx, attention_mask=torch.ones(1, S, dtype=torch.long),
                                  past_key_values=None, position_ids=None)
        l0 = lm.layers[0]
        st.update({
            "lm_x": bf16_bits(lx[0]),
            "lm_norm": bf16_bits(l0.input_layernorm(lx)[0]),
            "lm_attn": bf16_bits(l0.self_attn(lx, attention_mask=mask, position_embeddings=pe)[0][0]),
            "lm_mlp": bf16_bits(l0.mlp(lx)[0]),
            "lm_layer": bf16_bits(l0(lx, attention_mask=mask, position_embeddings=pe)[0]),
        })
    np.savez_compressed(
        os.path.join(OUT, "hf_tiny.npz"), grids=np.array(grids), ids=ids, pos3=pos3[:, 0].numpy(),
        pix=bf16_bits(pix), pooler=bf16_bits(vis.pooler_output), vit_last=bf16_bits(vis.last_hidden_state),
        logits=bf16_bits(out.logits[0]), **st)
    # greedy continuation with HF's own generate-equivalent loop (use_cache) for 8 tokens
    print("hf_tiny: S=%d, logits absmax %.3f" % (S, out.logits.float().abs().max()))


def gen_hf_truedim():
    """One ViT block (window + full) and one LM layer (prefill + 1 decode step) at the true 3B dimensions."""
    cfg = M.config_3b()
    cfg.vision.depth, cfg.text.num_hidden_layers = 1, 1
    cfg.text.vocab_size = 4096  # lm_head slice: first 4096 rows of the tied embedding
    W = WG.LazyWeights(cfg, seed=0)
    model = hf_model(cfg, W)
    grids = [(1, 16, 16)]
    N = 256
    widx, cu_win = M.vision_window_index(grids, 2, 112, 14)
    cos, sin = M.vit_rotary_tables(cfg.vision, grids, widx)
    x_in = synth_input("true.vit_x", (N, 1280)).to(torch.bfloat16)
    S = 40
    lx = synth_input("true.lm_x", (1, S + 1, 2048)).to(torch.bfloat16)
    pos3 = torch.arange(S + 1).view(1, 1, -1).expand(3, 1, -1).clone()
    pos3[1, 0, 10:30] += 3  # make the three axes differ
    pos3[2, 0, 10:30] += 5
    with torch.no_grad():
        blk = model.model.visual.blocks[0]
        win = blk(x_in, cu_seqlens=cu_win.to(torch.int32), position_embeddings=(cos, sin))
        full = blk(x_in, cu_seqlens=M.vision_full_seqlens(grids).to(torch.int32), position_embeddings=(cos, sin))
        lm = model.model.language_model
        out = lm(inputs_embeds=lx[:, :S], position_ids=pos3[:, :, :S], attention_mask=torch.ones(1, S, dtype=torch.long),
                 use_cache=True)
        out2 = lm(inputs_embeds=lx[:, S:], position_ids=pos3[:, :, S:], attention_mask=torch.ones(1, S + 1, dtype=torch.long),
                  past_key_values=out.past_key_values, use_cache=True)
        logits = model.lm_head(out2.last_hidden_state)[0, -1]
    np.savez_compressed(os.path.join(OUT, "hf_truedim.npz"), vit_x=bf16_bits(x_in), vit_block_win=bf16_bits(win),
                        vit_block_full=bf16_bits(full), lm_x=bf16_bits(lx[0]), pos3=pos3[:, 0].numpy(),
                        lm_prefill_hidden=bf16_bits(out.last_hidden_state[0]),
                        lm_decode_hidden=bf16_bits(out2.last_hidden_state[0]), lm_decode_logits=bf16_bits(logits))
    print("hf_truedim done")


def gen_index(ref_rope):
    from transformers.vision_utils import get_vision_position_ids, get_vision_window_index
    out = {}
    for name, grid in {"g32": [(1, 32, 32)], "g54": [(1, 54, 54)], "g64": [(1, 64, 64)], "gmix": [(1, 12, 8), (1, 8, 8)],
                       "g2x32": [(1, 32, 32), (1, 32, 32)]}.items():
        g = torch.tensor(grid)
        wi, cu = get_vision_window_index(g, spatial_merge_size=2, window_size=112, patch_size=14)
        out[name + "_grid"] = np.array(grid)
        out[name + "_window_index"] = wi.numpy()
        out[name + "_cu_window"] = cu.numpy().astype(np.int64)
        out[name + "_pos"] = get_vision_position_ids(g, 2).numpy()
    # mRoPE ids: 1- and 2-image prompts, left padded batch (reference function executed here)
    cfg = M.config_3b()
    rows, grids = [], []
    for k, gl in enumerate([[(1, 32, 32)], [(1, 32, 32), (1, 32, 32)], [(1, 12, 8)]]):
        rows.append(build_prompt(cfg, gl, 5 + k, 2, 11 - k, seed=20 + k))
        grids += gl
    L = max(len(r_) for r_ in rows) + 3
    ids = np.full((3, L), 151643, dtype=np.int64)
    am = np.zeros((3, L), dtype=np.int64)
    for i, r_ in enumerate(rows):
        ids[i, L - len(r_):] = r_
        am[i, L - len(r_):] = 1
    pos3, deltas = ref_rope(cfg, torch.tensor(ids), torch.tensor(grids), torch.tensor(am))
    out.update(rope_ids=ids, rope_mask=am, rope_grids=np.array(grids), rope_pos3=pos3.numpy(), rope_deltas=deltas.numpy())
    # text-only branch
    p2, d2 = ref_rope(cfg, torch.tensor(ids), None, torch.tensor(am))
    out.update(rope_text_pos3=p2.numpy(), rope_text_deltas=d2.numpy())
    np.savez_compressed(os.path.join(OUT, "index.npz"), **out)
    print("index done")


def gen_patchify():
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil, smart_resize
    from PIL import Image
    proc = Qwen2VLImageProcessorPil(min_pixels=56 * 56, max_pixels=768 * 768)
    out = {}
    for name, (h, w) in {"s56x84": (56, 84), "s448": (448, 448), "s756": (756, 756)}.items():
        rng = np.random.default_rng(1000 + h)
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        res = proc(images=[Image.fromarray(img)], return_tensors="np")
        pv = res["pixel_values"].astype(np.float32)
        out[name + "_seed"] = np.array([1000 + h, h, w])
        out[name + "_grid"] = res["image_grid_thw"][0]
        out[name + "_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(pv).tobytes()).digest(), dtype=np.uint8)
        out[name + "_head"] = pv[:6]
        if h * w < 10000:
            out[name + "_full"] = pv
    sr = [(h, w, *smart_resize(h, w, factor=28, min_pixels=56 * 56, max_pixels=768 * 768))
          for h, w in [(448, 448), (896, 896), (768, 768), (100, 37), (30, 30), (1000, 333), (756, 756), (57, 2000)]]
    out["smart_resize"] = np.array(sr)
    np.savez_compressed(os.path.join(OUT, "patchify.npz"), **out)
    print("patchify done")


def gen_reference_python():
    """Parsers, giou, output-layout functions: outputs of the reference's own code."""
    ns = {"re": __import__("re"), "json": json, "np": np, "torch": torch, "List": list, "Dict": dict, "Any": object}
    import typing
    ns.update(List=typing.List, Dict=typing.Dict, Any=typing.Any)
    extract("roll/pipeline/multi_utils.py", ["parse_points_text_from_content"], ns)
    extract("roll/pipeline/rlvr/seg_worker.py", ["parse_visual_prompt_from_json_s2"], ns)
    extract("roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py", ["compute_giou"], ns)
    cases = [
        '<think>t</think><answer>[{"bbox_2d": [10,100,200,210], "points": [[70,180],[20,200]]}]</answer>',
        '<answer>[{"bbox_2d": [1,2,3,4]}, {"bbox_2d": [5,6,7,8], "points": [[1,1]]}]</answer>',
        '<answer> [ {"bbox_2d": [1,2,3]} , {"bbox_2d": [5,6,7,8], "points": [[9,9,1]]} ] </answer>',
        'no tags at all',
        '<answer>not json</answer>',
        '<answer>{"bbox_2d": [1,2,3,4]}</answer>',
        '<answer>[1, "x", {"bbox_2d": [1,2,3,4], "points": []}]</answer><answer>[{"bbox_2d":[9,9,9,9]}]</answer>',
        '<answer>\n[{"bbox_2d": [1.5, 2, 3, 4], "points": [[0.5, 1]]}]\n</answer> trailing',
        '<answer><answer>[{"bbox_2d": [1,2,3,4]}]</answer></answer>',
        '<answer>[{"bbox_2d": [1,2,3,4], "points": [[1]]}]</answer>',
        '<answer>[{"bbox_2d": "abcd", "points": [[1,2]]}]</answer>',
        '<answer></answer>',
    ]
    import contextlib, io
    res = []
    for c in cases:
        with contextlib.redirect_stdout(io.StringIO()):
            res.append({"content": c, "points_text": ns["parse_points_text_from_content"](c),
                        "prompts": ns["parse_visual_prompt_from_json_s2"](c)})
    json.dump(res, open(os.path.join(OUT, "parsers.json"), "w"), indent=1)

    # raster: compute_giou from the reference; render through PIL itself
    from PIL import Image, ImageDraw
    rng = np.random.default_rng(3000)
    ras = {}
    for k in range(3):
        masks = np.zeros((4, 756, 756), dtype=np.uint8)
        for j in range(4):
            x0, y0 = rng.integers(0, 600, 2)
            ww, hh = rng.integers(20, 150, 2)
            masks[j, y0:y0 + hh, x0:x0 + ww] = 1
        gt = (rng.random((768, 768)) > 0.6).astype(np.uint8) * 255
        ras[f"rects{k}"] = np.array([[0, 0, 0, 0]])  # placeholder, masks regenerated from seed in the test
        acc = np.zeros((756, 756), dtype=np.uint8)
        for m in masks:
            acc = np.logical_or(acc, m).astype(np.uint8)  # seg_strategy.py:60 expression
        ras[f"union_sha{k}"] = np.frombuffer(hashlib.sha256(acc.tobytes()).digest(), dtype=np.uint8)
        ras[f"masks_sha{k}"] = np.frombuffer(hashlib.sha256(masks.tobytes()).digest(), dtype=np.uint8)
        ras[f"gt_sha{k}"] = np.frombuffer(hashlib.sha256(gt.tobytes()).digest(), dtype=np.uint8)
        # nearest 756->768 by the documented rule (cv2 absent) -- stored so the rule itself is frozen
        ys = np.minimum(np.floor(np.arange(768) * (1.0 / (768 / 756))).astype(np.int64), 755)   # OpenCV forms the inverse scale as a reciprocal
        up = acc[ys][:, ys]
        ras[f"resized_sha{k}"] = np.frombuffer(hashlib.sha256(up.tobytes()).digest(), dtype=np.uint8)
        ras[f"giou{k}"] = np.array(ns["compute_giou"](up, gt), dtype=np.float64)
        ras[f"counts{k}"] = np.array([np.logical_and(up > 0, gt > 0).sum(), np.logical_or(up > 0, gt > 0).sum()])
        # render: PIL does the work
        img = rng.integers(0, 256, (448, 448, 3), dtype=np.uint8)
        bbs = [[int(v) for v in rng.integers(0, 447, 4)] for _ in range(3)] + [[-5, -5, 30, 40], [400, 400, 500, 500]]
        bbs = [[min(b[0], b[2]), min(b[1], b[3]), max(b[0], b[2]), max(b[1], b[3])] for b in bbs] + [[50, 50, 40, 60]]
        pim = Image.fromarray(img).convert("RGBA")
        d = ImageDraw.Draw(pim)
        for b in bbs:
            try:
                d.rectangle([(b[0], b[1]), (b[2], b[3])], outline="blue", width=2)
            except Exception:
                continue
        ys2 = np.minimum(np.floor(np.arange(448) * (1.0 / (448 / 768))).astype(np.int64), 767)
        mk = up[ys2][:, ys2] > 0
        ov = np.zeros((448, 448, 4), dtype=np.uint8)
        ov[mk] = [255, 0, 0, int(255 * 0.4)]
        rendered = np.array(Image.alpha_composite(pim, Image.fromarray(ov, "RGBA")).convert("RGB"))
        ras[f"render_boxes{k}"] = np.array(bbs)
        ras[f"render_sha{k}"] = np.frombuffer(hashlib.sha256(rendered.tobytes()).digest(), dtype=np.uint8)
        ras[f"img_sha{k}"] = np.frombuffer(hashlib.sha256(img.tobytes()).digest(), dtype=np.uint8)
    ras["giou_empty"] = np.array(ns["compute_giou"](np.zeros((4, 4)), np.zeros((4, 4))))
    np.savez_compressed(os.path.join(OUT, "raster.npz"), **ras)

    # output layout: postprocess_generate & friends (TensorDict / DataProto stand-ins, this process only)
    class _DP:
        def __init__(self, batch=None, **kw):
            self.batch = batch
    for mod in ["roll", "roll.distributed", "roll.distributed.scheduler", "roll.distributed.scheduler.protocol"]:
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["roll.distributed.scheduler.protocol"].DataProto = _DP
    ns2 = {"torch": torch, "np": np, "TensorDict": lambda d, batch_size=None: d, "enum": __import__("enum")}
    extract("roll/utils/functionals.py", ["pad_to_length", "get_pad_mask", "concatenate_input_and_output",
                                          "postprocess_generate"], ns2)
    P, RL, SEQ, PAD, EOS = 12, 6, 20, 0, 2
    ids = torch.tensor([[0, 0, 0, 0, 5, 6, 7, 8, 9, 10, 11, 12],
                        [0, 0, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30],
                        [41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52]])
    am = (ids != 0).long()
    pos = torch.stack([torch.stack([(am[i].cumsum(0) - 1).clamp(min=0) + k for k in range(3)]) for i in range(3)])
    pos[am.unsqueeze(1).expand(-1, 3, -1) == 0] = 1
    outs = torch.tensor([[61, 62, 2, 0, 0, 0], [71, 72, 73, 74, 75, 2], [81, 0, 0, 0, 0, 0]])
    cat = ns2["concatenate_input_and_output"](ids, outs, 1)

    class _Prompts:
        batch = {"input_ids": ids, "attention_mask": am, "position_ids": pos}
    res = ns2["postprocess_generate"](_Prompts, cat.clone(), 1, SEQ, EOS, PAD).batch
    np.savez_compressed(os.path.join(OUT, "postprocess.npz"), in_ids=ids.numpy(), in_mask=am.numpy(), in_pos=pos.numpy(),
                        outs=outs.numpy(), cat=cat.numpy(), seq=np.array([SEQ, EOS, PAD]),
                        **{"out_" + k: v.numpy() for k, v in res.items()})
    print("reference python fixtures done")


def gen_render_image():
    """The reference's own ``render_image`` (rlvr_socioseg_vlm_pipeline_infer.py:383-452) executed on model-output-like
    bbox strings: thin / float / reversed / malformed boxes, and an image pair of unequal size (the LANCZOS branch).
    cv2 is absent: its one call (INTER_NEAREST resize) is answered by the documented rule, in this process only."""
    import typing
    from PIL import Image, ImageDraw

    class _CV2:
        INTER_NEAREST = 0

        @staticmethod
        def resize(a, size, interpolation=0):
            w, h = size
            ys = np.minimum(np.floor(np.arange(h) * (1.0 / (h / a.shape[0]))).astype(np.int64), a.shape[0] - 1)     # cv::resize: 1 / (dsize / ssize)
            xs = np.minimum(np.floor(np.arange(w) * (1.0 / (w / a.shape[1]))).astype(np.int64), a.shape[1] - 1)
            return a[ys][:, xs]
    ns = {"np": np, "json": json, "Image": Image, "ImageDraw": ImageDraw, "cv2": _CV2, "List": typing.List, "Dict": typing.Dict,
          "Any": typing.Any, "Union": typing.Union}
    extract("roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py", ["render_image"], ns)
    fn = ns["render_image"]
    rng = np.random.default_rng(4100)
    texts = [
        '[{"bbox_2d": [10, 12, 30, 40]}, {"bbox_2d": [5, 5, 5, 5]}, {"bbox_2d": [20, 3, 21, 50]}, {"bbox_2d": [2, 30, 60, 31]}]',
        '[{"bbox_2d": [1.9, 2.2, 7.5, 3.1]}, {"bbox_2d": [12.7, 9.2, 12.9, 30.8]}, {"bbox_2d": [3.9, 3.9, 3.1, 8]}, {"bbox_2d": [-4.5, -3.5, 6.2, 6.9]}]',
        '[{"bbox_2d": [1, 2, null, 4]}, {"bbox_2d": [8, 8, 20, 20]}, {"bbox_2d": ["1", 2, 3, 4]}, {"bbox_2d": [[1, 2], 3, 4, 5]}, {"bbox_2d": "abcd"}]',
        '[{"bbox_2d": [30, 30, 20, 40]}, {"bbox_2d": [30, 30, 40, 20]}, {"bbox_2d": [true, false, 9, 9]}, {"bbox_2d": [50, 60, 500, 600]}, {"bbox_2d": [-50, -60, 3, 2]}]',
        '[{"bbox_2d": [4, 4, 9, 9]}, {"bbox_2d": 5}]',
        '[{"bbox_2d": [4, 4, 9, 9]}, 7, {"points": [1, 2]}, {"bbox_2d": [1, 2, 3]}, {"bbox_2d": [14, 4, 14, 7]}, {"bbox_2d": [24, 4, 24, 6]}, {"bbox_2d": [34, 4, 36, 4]}]',
        'not json at all', '{"bbox_2d": [1, 2, 3, 4]}', '[]',
        '[{"bbox_2d": [1e10, 1, 2e10, 30]}, {"bbox_2d": [0, 0, 0, 0]}, {"bbox_2d": [63, 79, 63, 79]}, {"bbox_2d": [0.99, 0.99, 1.01, 1.01]}]',
    ]
    cases = []
    for k, txt in enumerate(texts + [None] * 6):
        h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        other = (h, w) if k % 3 else (int(rng.integers(30, 100)), int(rng.integers(30, 100)))
        if txt is None:              # random mixtures
            bb = []
            for _ in range(int(rng.integers(2, 9))):
                x0, y0 = float(rng.integers(-5, w + 3)), float(rng.integers(-5, h + 3))
                ww, hh = float(rng.integers(0, 4)) if rng.random() < 0.5 else float(rng.integers(0, 50)), \
                    float(rng.integers(0, 4)) if rng.random() < 0.5 else float(rng.integers(0, 50))
                if rng.random() < 0.4:
                    x0, y0, ww, hh = x0 + float(rng.random()), y0 + float(rng.random()), ww + float(rng.random()), hh + float(rng.random())
                    bb.append({"bbox_2d": [round(x0, 3), round(y0, 3), round(x0 + ww, 3), round(y0 + hh, 3)]})
                else:
                    bb.append({"bbox_2d": [int(x0), int(y0), int(x0 + ww), int(y0 + hh)]})
            txt = json.dumps(bb)
        seeds = [int(rng.integers(1, 1 << 30)) for _ in range(3)]
        imgs = [np.random.default_rng(seeds[0]).integers(0, 256, (h, w, 3), dtype=np.uint8),
                np.random.default_rng(seeds[1]).integers(0, 256, (other[0], other[1], 3), dtype=np.uint8)]
        mask = (np.random.default_rng(seeds[2]).random((48, 48)) > 0.55).astype(np.uint8)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            out = fn(txt, [Image.fromarray(a) for a in imgs], mask)
        cases.append({"bboxes_json": txt, "sizes": [[h, w], list(other)], "seeds": seeds,
                      "sha256": [hashlib.sha256(np.array(o).tobytes()).hexdigest() for o in out]})
    json.dump(cases, open(os.path.join(OUT, "render_image.json"), "w"), indent=1)
    print("render_image done:", len(cases), "cases,", sum(c["sizes"][0] != c["sizes"][1] for c in cases), "with unequal sizes")


def gen_weight_sync():
    """Wire format of the trainer -> engine weight push: the reference's OWN packer (send_recv_utils.py:64-152 TensorBucket /
    SendBucketManager, executed here on the CPU) fed with a mixed-dtype tensor set whose members straddle bucket boundaries;
    stored: every bucket's bytes, its meta_infos (after the reference's meta_to_dict, as they cross the RPC boundary) and the
    comm-plan lookup results of get_dist_info_from_comm_plan (functionals.py:875-882)."""
    import typing
    ns = {"torch": torch, "Dict": typing.Dict, "List": typing.List, "Optional": typing.Optional}
    extract("roll/utils/send_recv_utils.py", ["get_tensor_size", "TensorBucket", "SendBucketManager"], ns)
    BS = 4096
    mgr = ns["SendBucketManager"].__new__(ns["SendBucketManager"])       # the constructor hard-wires device="cuda": build by hand
    mgr.bucket_size = BS
    mgr.bucket = ns["TensorBucket"](BS, device="cpu")
    g = torch.Generator().manual_seed(77)
    tensors = {
        "model.layers.0.self_attn.q_proj.weight": torch.randn(48, 40, generator=g).to(torch.bfloat16),
        "model.layers.0.self_attn.q_proj.bias": torch.randn(48, generator=g).to(torch.bfloat16),
        "model.norm.weight": torch.randn(1000, generator=g),
        "visual.blocks.0.attn.qkv.weight": torch.randn(7, 3, 5, generator=g).to(torch.float16),
        "model.embed_tokens.weight": torch.randn(300, 23, generator=g).to(torch.bfloat16),
        "tiny": torch.randn(1, generator=g),
        "lm_head.weight": torch.randn(129, 17, generator=g),
    }
    buckets, metas = [], []

    def keep(meta_infos, buffer, nbytes):
        m = {k: dict(v) for k, v in meta_infos.items()}
        ns["SendBucketManager"].meta_to_dict(m)
        metas.append({k: {"bucket_start": int(v["bucket_start"]), "tensor_start": int(v["tensor_start"]), "save_bytes": int(v["save_bytes"]),
                          "tensor_meta": {"shape": v["tensor_meta"]["shape"], "dtype": str(v["tensor_meta"]["dtype"])}} for k, v in m.items()})
        buckets.append(buffer[:nbytes].clone().numpy().view(np.uint8))
    for name, t in tensors.items():
        for meta_infos, buffer in mgr.push_tensor(t, name):
            keep(meta_infos, buffer, BS)
    meta_infos, buffer = mgr.pop_last_bucket()
    if meta_infos is not None:
        keep(meta_infos, buffer, mgr.bucket.write_index)
    ns2 = {}
    extract("roll/utils/functionals.py", ["get_dist_info_from_comm_plan"], ns2)
    plan = {"0": {"group_name": "model_update_a_0_to_b_(0,0)-(1,0)-(2,1)", "master_addr": "127.0.0.1", "master_port": 29999, "src_pp_rank": 0, "src_rank": 0,
                  "tgt_devices": [{"rank": 0, "device": {"rank": 0, "node_rank": 0, "gpu_rank": 1}}, {"rank": 1, "device": {"rank": 0, "node_rank": 0, "gpu_rank": 2}},
                                  {"rank": 2, "device": {"rank": 1, "node_rank": 0, "gpu_rank": 5}}]},
            "1": {"group_name": "model_update_a_1_to_b_(3,0)", "master_addr": "127.0.0.1", "master_port": 29998, "src_pp_rank": 0, "src_rank": 1,
                  "tgt_devices": [{"rank": 3, "device": {"rank": 0, "node_rank": 0, "gpu_rank": 7}}]}}
    lookups = []
    for rc, rw in [(0, 0), (1, 0), (2, 1), (2, 0), (3, 0), (4, 0)]:
        r, a = ns2["get_dist_info_from_comm_plan"](plan, rank_in_cluster=rc, rank_in_worker=rw)
        lookups.append({"rank_in_cluster": rc, "rank_in_worker": rw, "rank": r, "group_name": None if a is None else a["group_name"]})
    np.savez_compressed(os.path.join(OUT, "weight_sync.npz"), **{f"bucket{i}": b for i, b in enumerate(buckets)},
                        **{"tensor_" + k: v.contiguous().view(-1).view(torch.uint8).numpy() for k, v in tensors.items()})
    json.dump({"bucket_size": BS, "metas": metas, "tensors": {k: {"shape": list(v.shape), "dtype": str(v.dtype)} for k, v in tensors.items()},
               "comm_plan": plan, "lookups": lookups}, open(os.path.join(OUT, "weight_sync.json"), "w"), indent=1)
    print("weight_sync done:", len(buckets), "buckets")


def make_ref_rope():
    ns = {"torch": torch, "Optional": __import__("typing").Optional, "Tuple": __import__("typing").Tuple}
    extract("mcore_adapter/src/mcore_adapter/models/qwen2_5_vl/modeling_qwen2_5_vl.py", ["get_rope_index"], ns)
    fn = ns["get_rope_index"]

    def call(cfg, input_ids, grid, mask):
        fake = types.SimpleNamespace(config=types.SimpleNamespace(
            merge_size=2, image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id,
            vision_start_token_id=cfg.vision_start_token_id, tokens_per_second=2))
        return fn(fake, input_ids, grid, None, None, mask)
    return call


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    rope = make_ref_rope()
    gen_reference_python()
    gen_render_image()
    gen_weight_sync()
    gen_index(rope)
    gen_patchify()
    gen_hf_tiny(rope)
    gen_hf_truedim()
    print("fixtures:", {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))})
