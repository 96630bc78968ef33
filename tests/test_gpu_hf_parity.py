"""HIP path compared DIRECTLY with HF transformers outputs (tests/golden/hf_*.npz, generated in the build container by
tools/make_golden.py / tools/make_golden_full.py from the real HF modules in bf16 eager mode) -- no oracle in between.

What can be expected: the reference path stores every activation in bf16, so two CORRECT
implementations that differ only in float32 summation order drift apart by rounding flips; the drift is noise (zero mean,
rms growing ~ sqrt(depth)).  The fixtures record how far this repo's CPU oracle is from HF at full depth
(``*_oracle_*`` keys: logits rms 0.043 / max 0.21 / |mean| < 1e-4 on logits of |x|max 3.9) -- the HIP path must land in
the same band: every test asserts rms, |mean signed error| (bias) and max, and prints the numbers.
"""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import bits_to_f32

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stats(got: torch.Tensor, want: torch.Tensor):
    d = (got.float().cpu() - want.float().cpu()).flatten()
    return {"max": float(d.abs().max()), "rms": float(d.pow(2).mean().sqrt()), "bias": float(d.mean()), "ref_absmax": float(want.abs().max())}


def record(name, payload):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "hf_parity_r02.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[name] = payload
    json.dump(cur, open(path, "w"), indent=1)


def test_tiny_model_hip_vs_hf_outputs(golden_dir):
    """Tiny config end to end: ViT pooler and the logits at EVERY position against HF's own outputs."""
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.engine import Engine
    g = np.load(os.path.join(golden_dir, "hf_tiny.npz"))
    e = Engine(geometry_tiny(), max_patches=512, max_prefill_tokens=256, max_batch=1, max_ctx=192, max_new_tokens=16)
    e.load_synthetic_weights(seed=0)
    grids = [tuple(x) for x in g["grids"].tolist()]
    emb = e.vit_forward(bits_to_f32(g["pix"]).cuda(), grids)
    sp = stats(emb, bits_to_f32(g["pooler"]))
    assert sp["max"] <= 4 * 2 ** -8 * sp["ref_absmax"] and sp["rms"] <= 0.004 and abs(sp["bias"]) <= 2e-4, sp
    logits = e.forward_logits([g["ids"]], [g["pos3"]], emb)
    sl = stats(logits, bits_to_f32(g["logits"]))          # HF's logits are bf16 (ulp 0.0078 at |x| in [1, 2)): part of the error
    assert sl["max"] <= 0.04 and sl["rms"] <= 0.008 and abs(sl["bias"]) <= 5e-4, sl
    record("tiny", {"pooler": sp, "logits_all_positions": sl})
    e.close()


def test_truedim_lm_layer_hip_vs_hf_outputs(golden_dir):
    """One LM layer + final norm + LM-head slice at the true 3B dimensions against HF's outputs.  The fixture's inputs are
    hidden states, not tokens: they enter the engine as 'image features' of a prompt made of image placeholders only.
    Row S of a (S+1)-token forward is the function HF evaluated as one decode step on its KV cache."""
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    g = np.load(os.path.join(golden_dir, "hf_truedim.npz"))
    geom = geometry_3b()
    geom.vision.depth, geom.text.num_hidden_layers, geom.text.vocab_size = 1, 1, 4096
    geom.vision.fullatt_block_indexes = ()
    geom.image_token_id, geom.vision_start_token_id, geom.vision_end_token_id = 4000, 4001, 4002
    e = Engine(geom, max_patches=256, max_prefill_tokens=64, max_batch=1, max_ctx=64, max_new_tokens=4)
    e.load_synthetic_weights(seed=0)
    lx = bits_to_f32(g["lm_x"]).to(torch.bfloat16).cuda()          # [S + 1, 2048]
    S1 = lx.shape[0]
    ids = np.full(S1, 4000, dtype=np.int64)
    logits = e.forward_logits([ids], [g["pos3"]], lx)
    s = stats(logits[S1 - 1], bits_to_f32(g["lm_decode_logits"]))
    assert s["max"] <= 0.02 and s["rms"] <= 0.005 and abs(s["bias"]) <= 5e-4, s
    # the decode kernels (GEMV + decode attention over the cache) on the same function: prefill S rows, then feed row S's
    # embedding as the pending token is not possible through the C ABI (tokens only), so the KV path is compared through the
    # full-depth fixture below; here: prefill of the first S rows leaves the same cache HF's prefill left
    record("truedim_lm_layer", {"decode_position_logits": s})
    e.close()


@pytest.mark.parametrize("tag", ["tile448", "pair448"])
def test_full_depth_3b_hip_vs_hf(golden_dir, tag):
    """FULL depth (32 ViT blocks + 36 LM layers, vocabulary 151 936) on BASELINE.json's tile (configs[1]: one 448x448 image,
    S = 448) and on the reference-faithful 2-image sample (S = 706): uint8 tile -> patchify -> ViT -> prefill -> 15
    teacher-forced decode steps through the KV cache, every stage against HF's outputs."""
    from socioreasoner_amd import synthetic
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    g = np.load(os.path.join(golden_dir, "hf_full3b.npz"))
    G = int(g["g_new"][0])
    geom = geometry_3b()
    e = Engine(geom, max_patches=2048, max_prefill_tokens=1024, max_batch=1, max_ctx=768, max_new_tokens=G)
    e.load_synthetic_weights(seed=0)
    tiles = g[f"{tag}_tiles"].tolist()
    hw = int(g[f"{tag}_hw"][0])
    grid = (1, hw // 14, hw // 14)
    ids, pos3 = g[f"{tag}_ids"], g[f"{tag}_pos3"]
    assert np.array_equal(ids, synthetic.tile_prompt(geom, tiles[0], grid, n_images=len(tiles)))      # the bench's own prompt
    pix = torch.cat([e.patchify(torch.from_numpy(synthetic.tile_pixels(i, hw, hw)).cuda()) for i in tiles], dim=0)
    emb = e.vit_forward(pix, [grid] * len(tiles))
    res = {"pooler": stats(emb, bits_to_f32(g[f"{tag}_pooler"]))}
    oracle_p, oracle_l = g[f"{tag}_oracle_pooler"], g[f"{tag}_oracle_logits_last"]
    sp = res["pooler"]
    assert sp["rms"] <= 1.5 * oracle_p[1] and sp["max"] <= 2.5 * oracle_p[0] and abs(sp["bias"]) <= 1e-3, (sp, oracle_p.tolist())
    logits = e.prefill([ids], [pos3], emb, return_logits=True)
    hf_last = bits_to_f32(g[f"{tag}_logits_last"])
    sl = res["prefill_logits"] = stats(logits[0], hf_last)
    assert sl["rms"] <= 1.5 * oracle_l[1] and sl["max"] <= 2.0 * oracle_l[0] and abs(sl["bias"]) <= 2e-3, (sl, oracle_l.tolist())
    hf_tokens = g[f"{tag}_tokens"].tolist()
    margin0 = float(g[f"{tag}_first_margin"][0])
    agree, decided = int(int(logits[0].argmax()) == hf_tokens[0]), 1
    if margin0 > 2 * sl["max"]:
        assert int(logits[0].argmax()) == hf_tokens[0]
    # teacher-forced decode on HF's tokens: trace[k + 1] = logits after feeding token k
    forced = torch.tensor([hf_tokens], dtype=torch.int32)
    _, trace = e.decode(G, trace=True, forced=forced, use_graph=False)
    stride = int(g["stride"][0])
    top_idx, top_val, samp, margin = g[f"{tag}_top_idx"], g[f"{tag}_top_val"], g[f"{tag}_sample"], g[f"{tag}_margin"]
    per_step = []
    for k in range(G - 1):
        lg = trace[k + 1, 0].cpu()
        st = stats(lg[::stride], bits_to_f32(samp[k]))
        tt = stats(lg[torch.from_numpy(top_idx[k]).long()], torch.from_numpy(top_val[k]))
        per_step.append({"sample": st, "top32": tt, "hf_margin": float(margin[k]), "token_equal": int(lg.argmax()) == hf_tokens[k + 1]})
        assert st["rms"] <= 1.6 * oracle_l[1] and st["max"] <= 2.0 * oracle_l[0] and abs(st["bias"]) <= 3e-3, (k, st)
        assert tt["max"] <= 2.0 * oracle_l[0], (k, tt)
        decided += 1
        agree += int(per_step[-1]["token_equal"])
        if margin[k] > 2 * max(st["max"], tt["max"]):          # HF's top-2 margin clear of the noise: the token must match
            assert per_step[-1]["token_equal"], (k, margin[k])
    res["decode_steps"] = {"rms_max": max(p["sample"]["rms"] for p in per_step), "max_max": max(p["sample"]["max"] for p in per_step),
                           "abs_bias_max": max(abs(p["sample"]["bias"]) for p in per_step),
                           "top32_max": max(p["top32"]["max"] for p in per_step), "greedy_tokens_equal": f"{agree}/{decided}",
                           "oracle_tokens_equal": f"{int(g[f'{tag}_oracle_token_agree'].sum())}/{len(g[f'{tag}_oracle_token_agree'])}"}
    res["oracle_vs_hf"] = {"pooler": dict(zip(("max", "rms", "bias", "ref_absmax"), oracle_p.tolist())),
                           "prefill_logits": dict(zip(("max", "rms", "bias", "ref_absmax"), oracle_l.tolist()))}
    print(tag, json.dumps(res))
    record(f"full3b_{tag}", res)
    e.close()
