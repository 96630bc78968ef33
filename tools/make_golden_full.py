#!/usr/bin/env python3
"""Full-depth golden fixtures: HF ``Qwen2_5_VLForConditionalGeneration`` (transformers 5.15.0, bf16, eager attention) at the
FULL SocioReasoner-3B geometry (32 ViT blocks + 36 LM layers, vocabulary 151 936) on the synthetic weights (seed 0) and on
the exact tiles BASELINE.json's configs are benchmarked on.  Runs ONLY in the build container (needs ~40 GB of RAM and a
few minutes of CPU); stores inputs/outputs only.

Fixtures (tests/golden/hf_full3b.npz):
  tile448_*  : configs[1] tile 0 -- one 448x448 image, S = 448 (the bench tile)
  pair448_*  : the reference-faithful sample shape -- two 448x448 images (tiles 0 and 1), S = 706
                (/root/reference/roll/pipeline/rlvr/rlvr_socioseg_vlm_pipeline_infer.py:61-124: map + satellite image)
per sample:  ids, pos3, pooler (bf16 bits, ViT + merger output), logits_last (bf16 bits [V], prefill last position),
             tokens [G] (HF greedy, lowest index among maxima), and for every one of the G teacher-forced decode steps
             the top-32 (index, value) of HF's logits, a strided vocabulary sample (every 37th id) and the top-2 margin.
Also stored: the error of THIS REPO'S ORACLE (oracle/model_ref.py) against the same HF run at full depth
(oracle_* keys) -- the calibration of what "two correct bf16 implementations" differ by at this depth.

The decode positions follow the reference rule (max position + 1 + i on all three mRoPE axes,
/root/reference/roll/utils/functionals.py:816-818).

Usage: python tools/make_golden_full.py [--tiny]      (--tiny: same flow on the tiny config, for a quick self-check)
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import make_golden as MG  # noqa: E402
from oracle import host_ref as H  # noqa: E402
from oracle import model_ref as M  # noqa: E402
from oracle import weights as WG  # noqa: E402
from socioreasoner_amd import synthetic  # noqa: E402
from socioreasoner_amd.config import geometry_3b, geometry_tiny  # noqa: E402

G_NEW = 16
TOPK = 32
STRIDE = 37


def greedy(logits_f32: torch.Tensor) -> int:
    """lowest index among the maxima (torch.argmax on CPU returns the first maximum)."""
    return int(torch.argmax(logits_f32))


def err_stats(a: torch.Tensor, b: torch.Tensor):
    d = (a.float() - b.float()).flatten()
    return np.array([float(d.abs().max()), float(d.pow(2).mean().sqrt()), float(d.mean()), float(b.float().abs().max())], dtype=np.float64)


def run_sample(tag, model, cfg, geom, W, tiles, hw, rope, with_oracle, out):
    grid = (1, hw // 14, hw // 14)
    grids = [grid] * len(tiles)
    pvs = []
    for i in tiles:
        pv, g = H.patchify(synthetic.tile_pixels(i, hw, hw))
        assert tuple(g) == grid
        pvs.append(pv)
    pv = torch.from_numpy(np.concatenate(pvs, axis=0))
    ids = synthetic.tile_prompt(geom, tiles[0], grid, n_images=len(tiles))
    S = len(ids)
    pos3, _ = rope(cfg, torch.tensor(ids)[None], torch.tensor(grids), torch.ones(1, S, dtype=torch.long))
    t0 = time.time()
    toks, tops_i, tops_v, samp, margin = [], [], [], [], []
    with torch.no_grad():
        vis = model.model.visual(pv.to(torch.bfloat16), torch.tensor(grids))
        pooler = vis.pooler_output
        o = model(input_ids=torch.tensor(ids)[None], attention_mask=torch.ones(1, S, dtype=torch.long), position_ids=pos3,
                  pixel_values=pv.to(torch.bfloat16), image_grid_thw=torch.tensor(grids), use_cache=True)
        logits_last = o.logits[0, -1]
        pkv = o.past_key_values
        nxt = greedy(logits_last.float())
        base = int(pos3.max()) + 1
        for k in range(G_NEW):
            toks.append(nxt)
            p = torch.full((3, 1, 1), base + k, dtype=torch.long)
            o = model(input_ids=torch.tensor([[nxt]]), attention_mask=torch.ones(1, S + k + 1, dtype=torch.long), position_ids=p,
                      past_key_values=pkv, use_cache=True)
            pkv = o.past_key_values
            lg = o.logits[0, -1].float()
            tv, ti = torch.topk(lg, TOPK)
            tops_i.append(ti.numpy().astype(np.int32))
            tops_v.append(tv.numpy())
            samp.append(MG.bf16_bits(lg[::STRIDE]))
            margin.append(float(tv[0] - tv[1]))
            nxt = greedy(lg)
    print(f"{tag}: S={S} HF forward + {G_NEW} steps {time.time() - t0:.1f}s; |logit|max {float(logits_last.float().abs().max()):.3f}; "
          f"tokens {toks}; min top-2 margin {min(margin):.4f}", flush=True)
    out.update({
        f"{tag}_tiles": np.array(tiles), f"{tag}_hw": np.array([hw]), f"{tag}_ids": ids, f"{tag}_pos3": pos3[:, 0].numpy(),
        f"{tag}_pooler": MG.bf16_bits(pooler), f"{tag}_logits_last": MG.bf16_bits(logits_last),
        f"{tag}_tokens": np.array(toks, dtype=np.int32), f"{tag}_top_idx": np.stack(tops_i), f"{tag}_top_val": np.stack(tops_v),
        f"{tag}_sample": np.stack(samp), f"{tag}_margin": np.array(margin),
        f"{tag}_first_margin": np.array([float(torch.topk(logits_last.float(), 2).values.diff().abs())]),
    })
    if with_oracle:
        # the oracle at full depth on the same inputs, teacher-forced on HF's tokens: how far two correct implementations of
        # the bf16 graph are apart at this depth (the calibration of the GPU tests' bounds)
        t0 = time.time()
        with torch.no_grad():
            emb = M.vit_forward(W, cfg, pv, grids)
            st = {"pooler": err_stats(emb, pooler)}
            x = M.embed_with_images(W, cfg, torch.from_numpy(ids), emb)
            caches = M.new_caches(cfg)
            lg = M.lm_forward(W, cfg, x, pos3[:, 0], caches)[0]
            st["logits_last"] = err_stats(lg, logits_last)
            st["logits_last_vs_bf16round"] = err_stats(M.r(lg), logits_last)
            agree = [greedy(lg) == toks[0]]
            step_err = []
            for k in range(G_NEW):
                xx = W["model.embed_tokens.weight"][torch.tensor([toks[k]])]
                lgk = M.lm_forward(W, cfg, xx, torch.full((3, 1), base + k), caches)[0]
                ref_top = torch.from_numpy(tops_v[k])
                step_err.append(err_stats(lgk[torch.from_numpy(tops_i[k]).long()], ref_top))
                if k + 1 < G_NEW:
                    agree.append(greedy(lgk) == toks[k + 1])
        for k_, v_ in st.items():
            out[f"{tag}_oracle_{k_}"] = v_
        out[f"{tag}_oracle_step_err"] = np.stack(step_err)
        out[f"{tag}_oracle_token_agree"] = np.array(agree)
        print(f"{tag}: oracle {time.time() - t0:.1f}s  [max, rms, mean, ref absmax] pooler {st['pooler']}  logits {st['logits_last']}  "
              f"step max {np.stack(step_err)[:, 0].max():.4f}  tokens agree {sum(agree)}/{len(agree)}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    rope = MG.make_ref_rope()
    if args.tiny:
        cfg, geom, hw, name = M.config_tiny(), geometry_tiny(), 112, "hf_fulltiny_selfcheck.npz"
    else:
        cfg, geom, hw, name = M.config_3b(), geometry_3b(), 448, "hf_full3b.npz"
    W = WG.LazyWeights(cfg, seed=0)
    t0 = time.time()
    model = MG.hf_model(cfg, W)
    print(f"HF model built in {time.time() - t0:.1f}s", flush=True)
    out = {"g_new": np.array([G_NEW]), "stride": np.array([STRIDE])}
    run_sample("tile448", model, cfg, geom, W, [0], hw, rope, not args.no_oracle, out)
    run_sample("pair448", model, cfg, geom, W, [0, 1], hw, rope, not args.no_oracle, out)
    path = os.path.join(MG.OUT, name)
    if args.tiny:
        path = os.path.join("/tmp", name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
