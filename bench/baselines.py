"""`cpu_baseline`: the reference's eager CPU path restated (oracle/model_ref.py, kind "port") and HF transformers' own bf16 module timed on this host's
cores, on a BOUNDED sample of the same workload.  This module, tests/ and __graft_entry__.smoke() are the only places that import oracle/ -- as the thing
measured BESIDE the product, never as part of it."""
from __future__ import annotations

import ctypes as C
import time

import torch

from .common import N_NEW

GRID = (1, 32, 32)


def device_weight_source(dev):
    """Synthetic parameters from the device generator (bit-identical to oracle/weights.py, pinned by test_synth_fill_bit_exact) -> host float32:
    seconds instead of minutes of single-threaded numpy."""
    from socioreasoner_amd import lib as L
    lib = L.load()
    s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def dev_weight(name, shape, base):
        n = 1
        for d in shape:
            n *= int(d)
        t = torch.empty(n, dtype=torch.bfloat16, device=dev)
        L.check(lib.sr_synth_fill(C.c_void_p(t.data_ptr()), n, name.encode(), 0, C.c_float(base), s), None, "sr_synth_fill")
        return t.float().cpu().reshape(tuple(shape))
    return dev_weight


def cpu_baseline(weight_source=None):
    """The oracle (a port of the reference's HF-eager CPU path) on this host's cores, on the same synthetic tile: the ViT and
    the 448-token prefill at FULL depth (32 blocks, 36 layers), then 16 greedy decode steps at full depth through the KV
    cache, extrapolated linearly to the 127 decode steps of a tile (the only extrapolation)."""
    from oracle import host_ref as H
    from oracle import model_ref as MR
    from oracle import weights as WG
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.config import geometry_3b
    cfg = MR.config_3b()
    W = WG.LazyWeights(cfg, seed=0, fast=True, source=weight_source)
    for n, _, _ in WG.param_specs(cfg):
        W[n]                                  # materialise outside the timed region
    img = synthetic.tile_pixels(0)
    ids = synthetic.tile_prompt(geometry_3b(), 0, GRID)
    p3, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], [GRID], None)
    p3 = p3[:, 0]
    nd = 16
    with torch.no_grad():
        t0 = time.perf_counter()
        pv, _ = H.patchify(img)
        emb = MR.vit_forward(W, cfg, torch.from_numpy(pv), [GRID])
        t1 = time.perf_counter()
        caches = MR.new_caches(cfg)
        x = MR.embed_with_images(W, cfg, torch.from_numpy(ids), emb)
        lg = MR.lm_forward(W, cfg, x, p3, caches)[0]
        t2 = time.perf_counter()
        base = int(p3.max()) + 1
        for k in range(nd):
            xx = W["model.embed_tokens.weight"][torch.tensor([int(lg.argmax())])]
            lg = MR.lm_forward(W, cfg, xx, torch.full((3, 1), base + k), caches)[0]
        t3 = time.perf_counter()
    vit_s, prefill_s = t1 - t0, t2 - t1
    decode_s = (N_NEW - 1) * (t3 - t2) / nd
    total = vit_s + prefill_s + decode_s
    return {"value": round(1.0 / total, 5), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/model_ref.py (float32 math, HF bf16 rounding points) on one synthetic 448x448 tile at FULL depth: 32 ViT blocks, "
                      f"36-layer 448-token prefill, {nd} greedy decode steps measured (ms/step x 127 = the tile's decode time)",
            "seconds_per_tile": round(total, 2),
            "phases_s": {"vit": round(vit_s, 2), "prefill": round(prefill_s, 2), "decode_127_steps": round(decode_s, 2),
                         "decode_s_per_step_measured": round((t3 - t2) / nd, 3)}}


def cpu_baseline_hf_bf16():
    """Library-grade CPU number beside the port: HF transformers' own Qwen2_5_VLForConditionalGeneration --
    the module the reference's hf_infer strategy calls (/root/reference/roll/distributed/strategy/hf_strategy.py:49-94) -- in bf16
    with sdpa attention on this host's cores, 3B geometry, random weights (values do not matter for the time), the same tile shape:
    ViT + 448-token prefill + 16 decode steps through the KV cache (x 127 / 16 for a tile's decode)."""
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLConfig
    from transformers.models.qwen2_5_vl.modeling_qwen2_5_vl import Qwen2_5_VLForConditionalGeneration
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.config import geometry_3b
    g = geometry_3b()
    v, t = g.vision, g.text
    c = Qwen2_5_VLConfig(
        vision_config=dict(depth=v.depth, hidden_size=v.hidden_size, num_heads=v.num_heads, intermediate_size=v.intermediate_size,
                           patch_size=v.patch_size, temporal_patch_size=v.temporal_patch_size, spatial_merge_size=v.spatial_merge_size,
                           window_size=v.window_size, fullatt_block_indexes=list(v.fullatt_block_indexes), out_hidden_size=v.out_hidden_size,
                           hidden_act="silu"),
        text_config=dict(num_hidden_layers=t.num_hidden_layers, hidden_size=t.hidden_size, num_attention_heads=t.num_attention_heads,
                         num_key_value_heads=t.num_key_value_heads, intermediate_size=t.intermediate_size, vocab_size=t.vocab_size,
                         rms_norm_eps=t.rms_norm_eps, rope_parameters={"rope_type": "default", "rope_theta": t.rope_theta,
                                                                       "mrope_section": list(t.mrope_section)},
                         max_position_embeddings=32768, tie_word_embeddings=True, bos_token_id=None, eos_token_id=None),
        image_token_id=g.image_token_id, video_token_id=g.image_token_id + 1, vision_start_token_id=g.vision_start_token_id,
        vision_end_token_id=g.vision_end_token_id, tie_word_embeddings=True)
    c._attn_implementation = "sdpa"
    t0 = time.perf_counter()
    with torch.device("meta"):
        model = Qwen2_5_VLForConditionalGeneration(c)
    model = model.to(torch.bfloat16).to_empty(device="cpu").eval()
    with torch.no_grad():
        for i_, p_ in enumerate(model.parameters()):     # finite, non-zero values: CPU GEMM time does not depend on them, a random fill of 3.75 G
            p_.fill_(0.004 + 0.001 * (i_ % 7))            # elements would cost a minute of the run
        for m_ in model.modules():           # rotary inv_freq buffers were emptied with the rest
            if hasattr(m_, "inv_freq") and hasattr(m_, "original_inv_freq"):
                inv, _ = m_.compute_default_rope_parameters(m_.config)
                m_.inv_freq = inv.float()
                m_.original_inv_freq = inv.float().clone()
            elif hasattr(m_, "inv_freq") and hasattr(m_, "theta"):
                m_.inv_freq = (1.0 / (m_.theta ** (torch.arange(0, m_.dim, 2, dtype=torch.float) / m_.dim))).float()
    build_s = time.perf_counter() - t0
    from oracle import host_ref as H          # (patchify of the baseline's input only)
    pv, _ = H.patchify(synthetic.tile_pixels(0))
    pv = torch.from_numpy(pv).to(torch.bfloat16)
    ids = synthetic.tile_prompt(g, 0, GRID)
    p3, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], [GRID], None)
    S, nd = len(ids), 16
    grid_t = torch.tensor([list(GRID)])
    with torch.no_grad():
        t0 = time.perf_counter()
        o = model(input_ids=torch.from_numpy(ids)[None], attention_mask=torch.ones(1, S, dtype=torch.long), position_ids=p3,
                  pixel_values=pv, image_grid_thw=grid_t, use_cache=True)
        t1 = time.perf_counter()
        pkv, nxt, base = o.past_key_values, int(o.logits[0, -1].float().argmax()), int(p3.max()) + 1
        for k in range(nd):
            o = model(input_ids=torch.tensor([[nxt]]), attention_mask=torch.ones(1, S + k + 1, dtype=torch.long),
                      position_ids=torch.full((3, 1, 1), base + k, dtype=torch.long), past_key_values=pkv, use_cache=True)
            pkv, nxt = o.past_key_values, int(o.logits[0, -1].float().argmax())
        t2 = time.perf_counter()
    fwd_s, dec_s = t1 - t0, (N_NEW - 1) * (t2 - t1) / nd
    return {"value": round(1.0 / (fwd_s + dec_s), 5), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "torch-bf16",
            "sample": f"transformers Qwen2_5_VLForConditionalGeneration (what the reference's hf_infer strategy calls), bf16, sdpa, random weights, "
                      f"one synthetic "
                      f"448x448 tile at FULL depth: ViT + 448-token prefill in one forward, {nd} decode steps through the KV cache measured (x 127 / {nd})",
            "seconds_per_tile": round(fwd_s + dec_s, 2),
            "phases_s": {"vit_plus_prefill": round(fwd_s, 2), "decode_127_steps": round(dec_s, 2), "decode_s_per_step_measured": round((t2 - t1) / nd, 4)},
            "model_build_s": round(build_s, 1)}
