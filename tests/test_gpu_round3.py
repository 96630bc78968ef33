"""Round 3 GPU tests.

1. "As close to exact arithmetic as the reference is" (VERDICT round 2, next #3): tests/golden/hf_truth3b.npz holds, for the full-depth
   3B model on the bench tile, the reference-faithful pair, a 756 x 756 tile (reference default max_pixels) and configs[4]'s 896 x 896
   tile, the outputs of HF in bf16 (the reference's eager path, /root/reference/roll/distributed/strategy/hf_strategy.py:49-94) AND of
   HF in float32 on the same bf16-representable weights.  Asserted per stage: err(HIP -> float32) <= BAND x err(HF-bf16 -> float32),
   rms and max, on identical elements -- plus the direct comparison with HF-bf16 the round-2 tests make, at the new shapes.
2. k_attn_prefill2 (LDS-DMA ring, shared tiles) is bit-identical to k_attn_prefill.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import bits_to_f32, switch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Two CORRECT bf16 implementations do not sit at exactly the same distance from float32: the CPU oracle of this repo lands at
# 0.94 .. 1.09 x HF-bf16's rms error and 0.95 .. 1.25 x its max error over the four samples (tools/make_golden_truth.py prints both).
RMS_BAND, MAX_BAND = 1.15, 1.6


def err(got: torch.Tensor, want: torch.Tensor):
    d = (got.float().cpu() - want.float().cpu()).flatten()
    return {"max": float(d.abs().max()), "rms": float(d.pow(2).mean().sqrt()), "bias": float(d.mean())}


def record(name, payload):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "hf_truth_r03.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[name] = payload
    json.dump(cur, open(path, "w"), indent=1)


def in_band(hip, ref, what):
    assert hip["rms"] <= RMS_BAND * ref["rms"] and hip["max"] <= MAX_BAND * ref["max"] and abs(hip["bias"]) <= 3e-3, (what, hip, ref)
    return {"hip_vs_f32": hip, "hf_bf16_vs_f32": ref, "rms_ratio": hip["rms"] / ref["rms"], "max_ratio": hip["max"] / ref["max"]}


@pytest.mark.parametrize("tag", ["tile448", "pair448", "tile756", "tile896"])
def test_full_depth_3b_distance_to_float32_truth(golden_dir, tag):
    from socioreasoner_amd import synthetic
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    g = np.load(os.path.join(golden_dir, "hf_truth3b.npz"))
    G, stride, ps, ls = int(g["g_new"][0]), int(g["stride"][0]), int(g["pool_stride"][0]), int(g["last_f32_stride"][0])
    geom = geometry_3b()
    tiles, hw = g[f"{tag}_tiles"].tolist(), int(g[f"{tag}_hw"][0])
    grid = (1, hw // 14, hw // 14)
    ids, pos3 = g[f"{tag}_ids"], g[f"{tag}_pos3"]
    S, N = len(ids), len(tiles) * grid[1] * grid[2]
    e = Engine(geom, max_patches=N, max_prefill_tokens=(S + 63) // 64 * 64, max_batch=1, max_ctx=(S + G + 64) // 64 * 64, max_new_tokens=G)
    e.load_synthetic_weights(seed=0)
    assert np.array_equal(ids, synthetic.tile_prompt(geom, tiles[0], grid, n_images=len(tiles)))
    pix = torch.cat([e.patchify(torch.from_numpy(synthetic.tile_pixels(i, hw, hw)).cuda()) for i in tiles], dim=0)
    emb = e.vit_forward(pix, [grid] * len(tiles))
    res = {}
    # ---- ViT + merger
    p16, p32 = bits_to_f32(g[f"{tag}_pooler"]), torch.from_numpy(g[f"{tag}_pooler_f32"])
    mine = emb.float().cpu().flatten()[::ps]
    res["pooler"] = in_band(err(mine, p32), err(p16, p32), "pooler")
    res["pooler"]["hip_vs_hf_bf16"] = err(mine, p16)
    # ---- prefill, last position
    logits = e.prefill([ids], [pos3], emb, return_logits=True)[0].cpu()
    l16, l32 = bits_to_f32(g[f"{tag}_logits_last"]), torch.from_numpy(g[f"{tag}_logits_last_f32"])
    res["prefill_logits"] = in_band(err(logits[::ls], l32), err(l16[::ls], l32), "prefill logits")
    res["prefill_logits"]["hip_vs_hf_bf16"] = d16 = err(logits, l16)
    ocal = g[f"{tag}_oracle_logits_last"]            # this repo's CPU oracle against HF-bf16: [max, rms, mean, absmax]
    assert d16["rms"] <= 1.5 * ocal[1] and d16["max"] <= 2.0 * ocal[0], (d16, ocal.tolist())
    hf_tokens = g[f"{tag}_tokens"].tolist()
    if float(g[f"{tag}_first_margin"][0]) > 2 * d16["max"]:
        assert int(logits.argmax()) == hf_tokens[0]
    # ---- teacher-forced decode through the KV cache
    _, trace = e.decode(G, trace=True, forced=torch.tensor([hf_tokens], dtype=torch.int32), use_graph=False)
    s16, s32 = g[f"{tag}_sample"], g[f"{tag}_sample_f32"]
    t_idx, t16, t32, margin = g[f"{tag}_top_idx"], g[f"{tag}_top_val"], g[f"{tag}_top_val_f32"], g[f"{tag}_margin"]
    steps, agree = [], 0
    for k in range(G - 1):
        lg = trace[k + 1, 0].cpu()
        st = in_band(err(lg[::stride], torch.from_numpy(s32[k])), err(bits_to_f32(s16[k]), torch.from_numpy(s32[k])), f"step {k}")
        top = lg[torch.from_numpy(t_idx[k]).long()]
        e_top, e_top16 = err(top, torch.from_numpy(t32[k])), err(torch.from_numpy(t16[k]), torch.from_numpy(t32[k]))
        # 32 values are too few for a ratio of maxima: the top-32 logits stay inside the band the strided sample set
        assert e_top["max"] <= MAX_BAND * max(e_top16["max"], st["hf_bf16_vs_f32"]["max"]), (k, e_top, e_top16)
        tok_eq = int(lg.argmax()) == hf_tokens[k + 1]
        agree += tok_eq
        if margin[k] > 2 * max(st["hip_vs_f32"]["max"], err(lg[::stride], bits_to_f32(s16[k]))["max"]):
            assert tok_eq, (k, float(margin[k]))
        steps.append(st)
    res["decode_steps"] = {"rms_ratio_max": max(s["rms_ratio"] for s in steps), "max_ratio_max": max(s["max_ratio"] for s in steps),
                           "hip_rms_max": max(s["hip_vs_f32"]["rms"] for s in steps), "hf_rms_max": max(s["hf_bf16_vs_f32"]["rms"] for s in steps),
                           "greedy_tokens_equal_hf": f"{agree}/{G - 1}"}
    print(tag, json.dumps(res))
    record(tag, res)
    e.close()


def _run_prefill(e, geom, tiles, hw, monkeypatch, attn2):
    from socioreasoner_amd import hostops, synthetic
    switch(monkeypatch, "SR_ATTN2", attn2)
    grid = (1, hw // 14, hw // 14)
    pix = torch.cat([e.patchify(torch.from_numpy(synthetic.tile_pixels(i, hw, hw)).cuda()) for i in tiles], dim=0)
    emb = e.vit_forward(pix, [grid] * len(tiles))
    ids, p3 = [], []
    for i in tiles:
        x = synthetic.tile_prompt(geom, i, grid, n_pre=11 + 7 * i, n_post=5 + 3 * i)       # ragged prompts
        pos3, _ = hostops.get_rope_index(torch.from_numpy(x)[None], [grid], None, image_token_id=geom.image_token_id,
                                         vision_start_token_id=geom.vision_start_token_id)
        ids.append(x)
        p3.append(pos3[:, 0].numpy())
    logits = e.prefill(ids, p3, emb, return_logits=True)
    toks = e.decode(4)
    torch.cuda.synchronize()
    return emb.clone(), logits.clone(), toks.clone()


@pytest.mark.parametrize("hw", [448, 896, 756])
def test_attention_v2_equals_round2_kernel_bit_for_bit(monkeypatch, hw):
    """ViT full-attention blocks (hd 80, 128-query blocks) and the causal GQA prefill (hd 128, 32 queries x 4 heads): the LDS-DMA
    kernel against the round-2 kernel on the same engine -- pooler, prefill logits and the first decoded tokens, torch.equal.
    448: four tiles (three images start off the first, ragged prompts, partial key tiles); 896: 4096 keys; 756: windows of 36..64
    patches and an image whose patch count is not a multiple of 64 (the second image starts 4 patches off a 16-byte boundary:
    the engine must keep that launch on the round-2 kernel)."""
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    geom = geometry_3b()
    geom.vision.depth, geom.text.num_hidden_layers = 8, 3                 # 7 windowed blocks + full-attention block 7; 3 LM layers
    geom.vision.fullatt_block_indexes = (7,)
    tiles = {448: [0, 1, 2, 3], 896: [0], 756: [0, 1]}[hw]
    n = (hw // 14) ** 2
    e = Engine(geom, max_patches=len(tiles) * n, max_prefill_tokens=len(tiles) * (n // 4 + 64), max_batch=len(tiles),
               max_ctx=(n // 4 + 64 + 63) // 64 * 64 + 64, max_new_tokens=4)
    e.load_synthetic_weights(seed=0)
    a = _run_prefill(e, geom, tiles, hw, monkeypatch, "0")
    b = _run_prefill(e, geom, tiles, hw, monkeypatch, "1")
    for x, y, what in zip(a, b, ("pooler", "prefill logits", "tokens")):
        assert torch.equal(x, y), (what, float((x.float() - y.float()).abs().max()))
    assert torch.isfinite(b[1]).all()
    e.close()


def test_abort_frees_the_row_at_once():
    """ABORT of a RUNNING request (reference vllm_strategy.py:188-193): the row stops between two decode chunks, the next poll hands row
    and KV slot to the waiting request, the aborted request reports nothing, the other requests decode exactly what they decode
    without the aborted neighbour."""
    from socioreasoner_amd import hostops, synthetic
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.engine import Engine
    from socioreasoner_amd.serving import ContinuousBatcher, Request
    geom = geometry_tiny()
    e = Engine(geom, max_patches=256, max_prefill_tokens=256, max_batch=2, max_ctx=192, max_new_tokens=64)
    e.load_synthetic_weights(seed=0)

    def req(i, max_new):
        ids = np.random.default_rng(50 + i).integers(0, 200, 20 + i).astype(np.int64)
        pos3, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], None, None)
        return Request(ids=ids, pos3=pos3[:, 0].numpy(), max_new=max_new, tag=i)

    def serve(abort_first):
        cb = ContinuousBatcher(e, eos=[], pad_id=0, steps_per_poll=4)
        a, b, c = req(0, 64), req(1, 64), req(2, 8)
        for r in (a, b, c):
            cb.submit(r)
        done, order = {}, []
        cb.pump(lambda r, t: (done.__setitem__(r.tag, t), order.append(r.tag)))     # a and b are running, c waits for a row
        assert len(cb.active) == 2 and len(cb.pending) == 1
        if abort_first:
            assert cb.abort(lambda r: r.tag == 0) == 1
        pumps = 0
        while not cb.idle():
            cb.pump(lambda r, t: (None if r.aborted else (done.__setitem__(r.tag, t), order.append(r.tag))))
            pumps += 1
            if abort_first and pumps == 1:
                assert 0 not in [r.tag for r in cb.active.values()], "the aborted row must be free after ONE poll"
        return done, order, pumps

    clean, order0, _ = serve(False)
    got, order1, pumps = serve(True)
    assert 0 not in got and got[1] == clean[1] and got[2] == clean[2]
    assert len(clean[1]) == 64 and len(clean[2]) == 8
    assert order1 == [2, 1], order1           # c took the aborted row and finished long before b; without the abort it waits for a row
    assert order0[-1] == 2
    e.close()


@pytest.mark.parametrize("mode", ["w8a16", "mx"])
def test_fp8_modes_distance_to_float32_truth_and_to_hf_bf16(golden_dir, mode):
    """configs[4]'s fp8 modes have no reference implementation (parity of the quantisers / kernels is against this repo's stated definition,
    DESIGN.md section 5).  This test puts them next to the one external yardstick there is: HF's float32 and bf16 runs of the SAME network with
    the SAME (bf16) weights, at full depth on the 448 tile.  The ViT is untouched by either mode (bit-identical pooler); the prefill logits
    move by the quantisation of the LM linears -- recorded as a multiple of HF-bf16's own distance to float32, bounded loosely (a broken
    scale or layout shows up as 10-100 x), and the greedy first token must survive wherever HF's top-2 margin exceeds the measured error."""
    from socioreasoner_amd import synthetic
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    g = np.load(os.path.join(golden_dir, "hf_truth3b.npz"))
    tag = "tile448"
    ps, ls = int(g["pool_stride"][0]), int(g["last_f32_stride"][0])
    geom = geometry_3b()
    tiles, hw = g[f"{tag}_tiles"].tolist(), int(g[f"{tag}_hw"][0])
    grid = (1, hw // 14, hw // 14)
    ids, pos3 = g[f"{tag}_ids"], g[f"{tag}_pos3"]
    S, N = len(ids), len(tiles) * grid[1] * grid[2]
    out = {}
    for name, fp8 in (("bf16", False), (mode, True if mode == "w8a16" else "mx")):
        e = Engine(geom, max_patches=N, max_prefill_tokens=(S + 63) // 64 * 64, max_batch=1, max_ctx=(S + 64) // 64 * 64, max_new_tokens=8, lm_fp8=fp8)
        e.load_synthetic_weights(seed=0)
        pix = torch.cat([e.patchify(torch.from_numpy(synthetic.tile_pixels(i, hw, hw)).cuda()) for i in tiles], dim=0)
        emb = e.vit_forward(pix, [grid] * len(tiles))
        logits = e.prefill([ids], [pos3], emb, return_logits=True)[0].cpu()
        out[name] = (emb.float().cpu().flatten()[::ps].clone(), logits.clone())
        e.close()
    assert torch.equal(out["bf16"][0], out[mode][0])                       # the vision tower does not see the LM's weight format
    l16, l32 = bits_to_f32(g[f"{tag}_logits_last"]), torch.from_numpy(g[f"{tag}_logits_last_f32"])
    ref = err(l16[::ls], l32)
    e_bf, e_q = err(out["bf16"][1][::ls], l32), err(out[mode][1][::ls], l32)
    q_vs_bf = err(out[mode][1], out["bf16"][1])
    res = {"hf_bf16_vs_f32": ref, "hip_bf16_vs_f32": e_bf, f"hip_{mode}_vs_f32": e_q, f"hip_{mode}_vs_hip_bf16": q_vs_bf,
           "rms_ratio_to_hf_bf16": e_q["rms"] / ref["rms"], "max_ratio_to_hf_bf16": e_q["max"] / ref["max"]}
    print(mode, json.dumps(res))
    record(f"fp8_{mode}_tile448", res)
    # measured (round 3, random N(0, 0.02) weights -- e4m3's 2^-4 relative step on every LM weight, 36 layers deep): rms / max ratio w8a16 8.0 / 8.5, mx 10.6 / 10.8, both unbiased
    assert e_q["rms"] <= 12.0 * ref["rms"] and e_q["max"] <= 12.0 * ref["max"] and abs(e_q["bias"]) <= 2e-2, res
    assert e_q["rms"] >= 0.9 * e_bf["rms"]                                  # (quantising cannot bring the result closer to float32 than bf16 weights do)
    margin = float(g[f"{tag}_first_margin"][0])
    if margin > 2 * max(e_q["max"], err(out[mode][1], l16)["max"]):
        assert int(out[mode][1].argmax()) == int(g[f"{tag}_tokens"][0])


@pytest.mark.parametrize("hw", [448, 896, 756])
def test_window_attention_without_lds_equals_the_staged_kernel_bit_for_bit(monkeypatch, hw):
    """k_attn_win64 (one wave per (window, head), K / V^T fragments straight from global memory) against k_attn_prefill<80> on the ViT's window
    blocks at full 3B depth: the vision embeddings must be identical.  756-pixel tiles have ragged border windows: the engine must keep the
    staged kernel there (same result by construction)."""
    from socioreasoner_amd import synthetic
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    geom = geometry_3b()
    grid = (1, hw // 14, hw // 14)
    n = 2
    e = Engine(geom, max_patches=n * grid[1] * grid[2], max_prefill_tokens=64, max_batch=1, max_ctx=128, max_new_tokens=8)
    e.load_synthetic_weights(seed=0)
    pix = torch.cat([e.patchify(torch.from_numpy(synthetic.tile_pixels(70 + i, hw, hw)).cuda()) for i in range(n)], dim=0)
    out = {}
    for mode in ("0", "1"):
        switch(monkeypatch, "SR_ATTN_WIN64", mode)
        out[mode] = e.vit_forward(pix, [grid] * n).clone()
        torch.cuda.synchronize()
    # (round 4: 448- and 896-pixel tiles -- grids that ARE a multiple of the window, where HF pads an empty window row / column -- used to be
    # refused the kernel because of those empty windows; the plan says which kernel a geometry takes)
    assert e.vit_plan()["windows_all_64"] == (hw in (448, 896)), e.vit_plan()
    assert torch.equal(out["0"].view(torch.int16), out["1"].view(torch.int16))
    e.close()
