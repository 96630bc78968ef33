"""CPU: the oracle against the golden vectors generated from the HF dependency and the reference's own
pure functions (tools/make_golden.py).  This is what pins oracle/ (see oracle/__init__.py)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import host_ref as H
from oracle import model_ref as M
from oracle import raster_ref as R
from oracle import weights as WG
from tests.util import assert_bf16_close, bits_to_f32, bf16_compare


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


@pytest.fixture(scope="module")
def tiny(golden_dir):
    g = np.load(os.path.join(golden_dir, "hf_tiny.npz"))
    cfg = M.config_tiny()
    return g, cfg, WG.LazyWeights(cfg, seed=0)


def test_single_ops_match_hf_bf16(tiny):
    """Each op on identical inputs: at most 1 bf16 ulp, <=1% of elements (accumulation-order flips only)."""
    g, cfg, W = tiny
    grids = [tuple(x) for x in g["grids"].tolist()]
    vc = cfg.vision
    x = bits_to_f32(g["vit_x"])
    widx, cu_win = M.vision_window_index(grids, 2, 112, 14)
    cos, sin = M.vit_rotary_tables(vc, grids, widx)
    p = "visual.blocks.0."
    assert_bf16_close(M.rmsnorm(x, W[p + "norm1.weight"], 1e-6), bits_to_f32(g["vit_norm1"]), 0, 0, "vit norm")
    assert_bf16_close(M.linear(x, W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"]), bits_to_f32(g["vit_qkv"]), what="qkv")
    assert_bf16_close(M.vit_attention(W, p, vc, x, cu_win, cos, sin), bits_to_f32(g["vit_attn_win"]), what="attn win")
    assert_bf16_close(M.vit_attention(W, p, vc, x, M.vision_full_seqlens(grids), cos, sin),
                      bits_to_f32(g["vit_attn_full"]), what="attn full")
    assert_bf16_close(M.vit_mlp(W, p, x), bits_to_f32(g["vit_mlp"]), what="vit mlp")
    assert_bf16_close(M.vit_block(W, 0, vc, x, cu_win, cos, sin), bits_to_f32(g["vit_block_win"]), 2, 0.02, "block")
    assert_bf16_close(M.vit_merger(W, vc, x), bits_to_f32(g["vit_merger"]), 2, 0.02, "merger")
    assert_bf16_close(M.linear(bits_to_f32(g["pix"]), W["visual.patch_embed.proj.weight"]),
                      bits_to_f32(g["vit_patch_embed"]), what="patch embed")
    tc = cfg.text
    lx = bits_to_f32(g["lm_x"])
    pos3 = torch.from_numpy(g["pos3"])
    c2, s2 = M.mrope_tables(tc, pos3)
    q = "model.layers.0."
    assert_bf16_close(M.rmsnorm(lx, W[q + "input_layernorm.weight"], tc.rms_norm_eps), bits_to_f32(g["lm_norm"]), 0, 0, "lm norm")
    assert_bf16_close(M.lm_attention(W, q, tc, lx, c2, s2, {}), bits_to_f32(g["lm_attn"]), what="lm attn")
    assert_bf16_close(M.lm_mlp(W, q, lx), bits_to_f32(g["lm_mlp"]), what="lm mlp")
    assert_bf16_close(M.lm_layer(W, 0, tc, lx, c2, s2, {}), bits_to_f32(g["lm_layer"]), 2, 0.02, "lm layer")


def test_end_to_end_within_bf16_noise_floor(tiny):
    """Whole tiny model vs HF bf16 eager.  Identical rounding points still leave the bf16 noise floor
    (DESIGN.md 'Numerics'): two correct implementations differ by ~1 bf16 ulp of the output."""
    g, cfg, W = tiny
    grids = [tuple(x) for x in g["grids"].tolist()]
    pix = bits_to_f32(g["pix"])
    img = M.vit_forward(W, cfg, pix, grids)
    want = bits_to_f32(g["pooler"])
    mu, frac, mad = bf16_compare(img, want)
    assert mad <= 2 * float(want.abs().max()) * 2 ** -8, (mu, frac, mad)  # 2 bf16 steps of the largest element
    ids = torch.from_numpy(g["ids"])
    x = M.embed_with_images(W, cfg, ids, img)
    lg = M.lm_forward(W, cfg, x, torch.from_numpy(g["pos3"]), M.new_caches(cfg), all_logits=True)
    hf = bits_to_f32(g["logits"])
    d = (lg - hf).abs()
    assert float(d.max()) <= 0.04 and float(d.pow(2).mean().sqrt()) <= 0.008, (float(d.max()), float(d.pow(2).mean().sqrt()))
    # greedy choice agrees wherever HF's own top-2 margin exceeds the noise
    top2 = hf.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.05
    assert (lg.argmax(-1)[clear] == hf.argmax(-1)[clear]).all()


def test_truedim_slices(golden_dir):
    g = np.load(os.path.join(golden_dir, "hf_truedim.npz"))
    cfg = M.config_3b()
    cfg.vision.depth, cfg.text.num_hidden_layers, cfg.text.vocab_size = 1, 1, 4096
    W = WG.LazyWeights(cfg, seed=0)
    grids = [(1, 16, 16)]
    widx, cu_win = M.vision_window_index(grids, 2, 112, 14)
    cos, sin = M.vit_rotary_tables(cfg.vision, grids, widx)
    x = bits_to_f32(g["vit_x"])
    assert_bf16_close(M.vit_block(W, 0, cfg.vision, x, cu_win, cos, sin), bits_to_f32(g["vit_block_win"]), 2, 0.08, "win")
    assert_bf16_close(M.vit_block(W, 0, cfg.vision, x, M.vision_full_seqlens(grids), cos, sin),
                      bits_to_f32(g["vit_block_full"]), 2, 0.08, "full")
    lx = bits_to_f32(g["lm_x"])
    pos3 = torch.from_numpy(g["pos3"])
    S = lx.shape[0] - 1
    tc = cfg.text
    caches = M.new_caches(cfg)
    c, s = M.mrope_tables(tc, pos3[:, :S])
    h = M.lm_layer(W, 0, tc, lx[:S], c, s, caches[0])
    hn = M.rmsnorm(h, W["model.norm.weight"], tc.rms_norm_eps)
    assert_bf16_close(hn, bits_to_f32(g["lm_prefill_hidden"]), 4, 0.30, "prefill hidden")
    c, s = M.mrope_tables(tc, pos3[:, S:])
    h2 = M.lm_layer(W, 0, tc, lx[S:], c, s, caches[0])
    hn2 = M.rmsnorm(h2, W["model.norm.weight"], tc.rms_norm_eps)
    assert_bf16_close(hn2, bits_to_f32(g["lm_decode_hidden"]), 4, 0.30, "decode hidden")
    logits = hn2 @ W["lm_head.weight"][:4096].t()
    want = bits_to_f32(g["lm_decode_logits"])
    assert float((M.r(logits[0]) - want).abs().max()) <= float(want.abs().max()) * 2 ** -7


def test_index_math(golden_dir):
    g = np.load(os.path.join(golden_dir, "index.npz"))
    for name in ["g32", "g54", "g64", "gmix", "g2x32"]:
        grid = [tuple(x) for x in g[name + "_grid"].tolist()]
        wi, cu = M.vision_window_index(grid, 2, 112, 14)
        assert (wi.numpy() == g[name + "_window_index"]).all(), name
        assert (cu.numpy() == g[name + "_cu_window"]).all(), name
        assert (M.vision_position_ids(grid, 2).numpy() == g[name + "_pos"]).all(), name
    pos3, deltas = H.get_rope_index(g["rope_ids"], g["rope_grids"], g["rope_mask"])
    assert (pos3 == g["rope_pos3"]).all() and (deltas == g["rope_deltas"]).all()
    p2, d2 = H.get_rope_index(g["rope_ids"], None, g["rope_mask"])
    assert (p2 == g["rope_text_pos3"]).all() and (d2 == g["rope_text_deltas"]).all()


def test_patchify_and_smart_resize(golden_dir):
    g = np.load(os.path.join(golden_dir, "patchify.npz"))
    for name in ["s56x84", "s448", "s756"]:
        seed, h, w = g[name + "_seed"].tolist()
        img = np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)
        pv, grid = H.patchify(img)
        assert tuple(g[name + "_grid"].tolist()) == grid
        assert (pv[:6] == g[name + "_head"]).all()
        assert (sha(pv) == g[name + "_sha256"]).all(), name
    for h, w, eh, ew in g["smart_resize"].tolist():
        assert H.smart_resize(h, w) == (eh, ew)
    lut = R.normalize_lut()
    x = np.arange(256, dtype=np.uint8)[:, None, None].repeat(3, 2)
    assert (H.normalize_u8(x)[:, 0, :].T == lut).all()


def test_parsers(golden_dir):
    for case in json.load(open(os.path.join(golden_dir, "parsers.json"))):
        assert H.parse_points_text_from_content(case["content"]) == case["points_text"]
        assert H.parse_visual_prompt_from_json_s2(case["content"]) == case["prompts"], case["content"]


def test_raster(golden_dir):
    g = np.load(os.path.join(golden_dir, "raster.npz"))
    rng = np.random.default_rng(3000)
    for k in range(3):
        masks = np.zeros((4, 756, 756), dtype=np.uint8)
        for j in range(4):
            x0, y0 = rng.integers(0, 600, 2)
            ww, hh = rng.integers(20, 150, 2)
            masks[j, y0:y0 + hh, x0:x0 + ww] = 1
        gt = (rng.random((768, 768)) > 0.6).astype(np.uint8) * 255
        assert (sha(masks) == g[f"masks_sha{k}"]).all() and (sha(gt) == g[f"gt_sha{k}"]).all()
        for impl in (H, R):
            acc = impl.mask_union(list(masks))
            assert (sha(acc) == g[f"union_sha{k}"]).all()
            up = impl.resize_nearest(acc, 768, 768)
            assert (sha(up) == g[f"resized_sha{k}"]).all()
            assert list(impl.iou_counts(up, gt)) == g[f"counts{k}"].tolist()
        assert H.compute_giou(up, gt) == float(g[f"giou{k}"])
        img = rng.integers(0, 256, (448, 448, 3), dtype=np.uint8)
        assert (sha(img) == g[f"img_sha{k}"]).all()
        _ = [[int(v) for v in rng.integers(0, 447, 4)] for _ in range(3)]  # keep the generator in step
        bbs = g[f"render_boxes{k}"].tolist()
        for impl in (H, R):
            assert (sha(impl.render_overlay(img, up, bbs)) == g[f"render_sha{k}"]).all(), (k, impl.__name__)
    assert H.compute_giou(np.zeros((4, 4)), np.zeros((4, 4))) == float(g["giou_empty"]) == 1.0


def test_nearest_rule_is_opencvs_reciprocal_form():
    """cv::resize forms the inverse scale as 1 / (dsize / ssize) and resizeNN takes floor(dx * that).  For most size pairs this
    equals floor(dx * ssize / dsize); 768 -> 1148 is one where it does not (2 of 1148 columns) -- both oracles follow OpenCV."""
    src = np.arange(768, dtype=np.int64)
    want = np.minimum(np.floor(np.arange(1148) * (1.0 / (1148 / 768))).astype(np.int64), 767)
    naive = np.minimum(np.floor(np.arange(1148) * (768 / 1148)).astype(np.int64), 767)
    assert int((want != naive).sum()) == 2
    img = (src[None, :] % 251).astype(np.uint8).repeat(4, axis=0)
    for impl in (H, R):
        assert np.array_equal(impl.resize_nearest(img, 4, 1148)[0], img[0][want]), impl.__name__
    for s_, d_ in ((756, 768), (768, 448), (768, 756), (768, 896)):        # the sizes on the shipped path: both forms agree
        a = np.minimum(np.floor(np.arange(d_) * (1.0 / (d_ / s_))).astype(np.int64), s_ - 1)
        assert np.array_equal(a, np.minimum(np.floor(np.arange(d_) * (s_ / d_)).astype(np.int64), s_ - 1))


def test_render_image_reference_function(golden_dir):
    """The reference's render_image executed on thin / float / reversed / malformed boxes and on image pairs of unequal
    size (tools/make_golden.py gen_render_image): the oracle reproduces every output image."""
    import json
    cases = json.load(open(os.path.join(golden_dir, "render_image.json")))
    assert sum(c["sizes"][0] != c["sizes"][1] for c in cases) >= 3
    for k, c in enumerate(cases):
        imgs = [np.random.default_rng(sd).integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8) for sd, hw in zip(c["seeds"][:2], c["sizes"])]
        mask = (np.random.default_rng(c["seeds"][2]).random((48, 48)) > 0.55).astype(np.uint8)
        got = H.render_image(c["bboxes_json"], imgs, mask)
        assert [hashlib.sha256(np.ascontiguousarray(g).tobytes()).hexdigest() for g in got] == c["sha256"], (k, c["bboxes_json"])


def test_postprocess_generate(golden_dir):
    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    seq, eos, pad = g["seq"].tolist()
    cat = H.concatenate_input_and_output(g["in_ids"], g["outs"], 1)
    assert (cat == g["cat"]).all()
    out = H.postprocess_generate(g["in_ids"], g["in_mask"], g["in_pos"], cat, 1, seq, eos, pad)
    for k, v in out.items():
        assert (np.asarray(v).astype(np.int64) == g["out_" + k].astype(np.int64)).all(), k


def test_weight_generator_properties():
    w = WG.synth_f32("model.layers.0.mlp.gate_proj.weight", (512, 256), seed=0)
    assert abs(float(w.std()) - 0.02) < 5e-4 and abs(float(w.mean())) < 2e-4
    assert (torch.from_numpy(w).bfloat16().float().numpy() == w).all()
    w2 = WG.synth_f32("model.layers.0.mlp.gate_proj.weight", (512 * 256,), seed=0, start=0)
    assert (w2 == w.reshape(-1)).all()
    assert (WG.synth_f32("a", (64,), seed=0) != WG.synth_f32("b", (64,), seed=0)).any()
    assert WG.tensor_key("model.norm.weight", 0) == WG.tensor_key("model.norm.weight", 0)


@pytest.mark.parametrize("tag", ["tiny", "large"])
def test_sam2_oracle_matches_hf_sam2_model(golden_dir, tag):
    """oracle/sam2_ref.py (float32) against HF Sam2Model's float32 outputs (tools/make_golden_sam2.py): stage outputs, FPN / decoder
    features, low-resolution mask logits, IoU scores, the selected 756 x 756 mask -- for a box, a box with clicks, and a single click."""
    from oracle import sam2_ref as S
    from socioreasoner_amd import synthetic
    g = np.load(os.path.join(golden_dir, "sam2.npz"))
    geom = S.geometry_tiny() if tag == "tiny" else S.geometry_large()
    W = S.synthetic_weights(geom)
    hw, st = int(g[f"{tag}_hw"][0]), int(g[f"stride_{tag}"][0])
    o = S.Sam2Oracle(W, geom)
    o.set_image(synthetic.tile_pixels(int(g[f"{tag}_img_seed"][0]), hw, hw))
    for i, x in enumerate(o.stages):
        want = g[f"{tag}_stage{i}_f32"]
        assert np.abs(x.flatten()[::st].numpy() - want).max() <= 2e-4 * max(1.0, np.abs(want).max()), i
    for i, x in enumerate(o.feats):
        want = g[f"{tag}_feat{i}_f32"]
        assert np.abs(x.flatten()[::st].numpy() - want).max() <= 2e-4 * max(1.0, np.abs(want).max()), i
    for p in range(3):
        box = g[f"{tag}_p{p}_box"].tolist() or None
        pts = g[f"{tag}_p{p}_pts"]
        masks, iou, low = o.predict(pts if len(pts) else None, g[f"{tag}_p{p}_labels"] if len(pts) else None, box)
        assert np.abs(low - g[f"{tag}_p{p}_low"]).max() <= 1e-3, p
        assert np.abs(iou - g[f"{tag}_p{p}_iou"]).max() <= 1e-5, p
        best = int(np.argmax(iou))
        assert best == int(g[f"{tag}_p{p}_best"][0])
        want = np.unpackbits(g[f"{tag}_p{p}_mask_bits"])[: hw * hw].reshape(hw, hw).astype(bool)
        assert (masks[best] != want).sum() <= 2, p              # (a logit within 1e-6 of zero may land on either side)


def test_sam2_predictor_contract_against_hf_processor(golden_dir):
    """What the sam2 package's predictor does AROUND the network, pinned to HF's independent implementation of the same contract
    (tests/golden/sam2_contract.npz, tools/make_golden_sam2_contract.py): (1) prompt scaling -- ``Sam2Processor``'s normalised points / boxes
    equal ``oracle.sam2_ref.prompt_points`` (the box as its two corners in front of the clicks); (2) box handling -- HF's NATIVE ``input_boxes``
    path (corners + 0.5, point_embed[2] / [3], padding point) gives the logits / IoUs the oracle computes from the box-as-labelled-points
    form; (3) post-processing -- ``Sam2ImageProcessor.post_process_masks`` (bilinear, align_corners False, > 0) equals
    ``oracle.sam2_ref.postprocess`` on the same logits, all three masks.  Still unpinned: SAM2Transforms' resize + normalisation (torchvision)
    and the sam2_hiera_large.pt name table."""
    from oracle import sam2_ref as S
    from socioreasoner_amd import synthetic
    g = np.load(os.path.join(golden_dir, "sam2_contract.npz"))
    hw, size = int(g["hw"][0]), int(g["image_size"][0])
    geom = S.geometry_tiny()
    assert geom.image_size == size
    o = S.Sam2Oracle(S.synthetic_weights(geom), geom)
    o.set_image(synthetic.tile_pixels(7, hw, hw))
    for p in range(3):
        box = g[f"p{p}_box"].tolist() or None
        pts = g[f"p{p}_pts"]
        labels = g[f"p{p}_labels"] if len(pts) else None
        c, l = S.prompt_points(box, pts if len(pts) else None, labels, (hw, hw), size)
        nb = 2 if box is not None else 0
        if box is not None:                                                    # (1) the processor's scaled box = the two corner points, labels 2 / 3
            assert np.array_equal(c[:2].numpy().reshape(-1), g[f"p{p}_hf_boxes"]) and l[:2].tolist() == [2, 3]
        assert np.array_equal(c[nb:].numpy(), g[f"p{p}_hf_points"].reshape(-1, 2)) and l[nb:].tolist() == (labels.tolist() if labels is not None else [])
        masks, iou, low = o.predict(pts if len(pts) else None, labels, box)   # (2) against HF's native box path
        assert np.abs(low - g[f"p{p}_low_native"]).max() <= 2e-4 and np.abs(iou - g[f"p{p}_iou_native"]).max() <= 1e-5, p
        _, m3, _ = S.postprocess(torch.from_numpy(g[f"p{p}_low_native"]), torch.from_numpy(g[f"p{p}_iou_native"]), (hw, hw))    # (3)
        want = np.unpackbits(g[f"p{p}_masks_bits"])[: 3 * hw * hw].reshape(3, hw, hw)
        assert np.array_equal(m3.numpy(), want), p
