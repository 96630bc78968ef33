"""SAM2 (Hiera-L) on the device against HF ``Sam2Model``'s own outputs (tests/golden/sam2.npz, tools/make_golden_sam2.py).

The reference runs SAM2 in float32; the device path stores bf16 (float32 accumulation).  The fixtures hold HF's float32 run AND HF's
bfloat16 run of the same network on the same weights: every stage is held to BAND x the distance HF-bf16 itself has from HF-float32,
and the final 756 x 756 masks must be exact wherever the float32 logit is clear of that distance."""
import json
import os

import numpy as np
import pytest
import torch

from tests.util import bits_to_f32

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAND = 1.6


def _err(a, b):
    d = (torch.as_tensor(a).float().flatten() - torch.as_tensor(b).float().flatten())
    return float(d.abs().max()), float(d.pow(2).mean().sqrt())


def _engine(tag, dtype=torch.bfloat16):
    from oracle import sam2_ref as S
    from socioreasoner_amd import sam2
    og = S.geometry_tiny() if tag == "tiny" else S.geometry_large()
    g = sam2.Sam2Geometry(**{k: getattr(og, k) for k in sam2.Sam2Geometry.__dataclass_fields__})
    e = sam2.Sam2Engine(g, dtype=dtype)
    e.load_state_dict(S.synthetic_weights(og))
    return e, og


@pytest.mark.parametrize("tag", ["tiny", "large"])
def test_sam2_device_vs_hf(golden_dir, tag):
    from oracle import sam2_ref as S
    from socioreasoner_amd import sam2, synthetic
    g = np.load(os.path.join(golden_dir, "sam2.npz"))
    e, og = _engine(tag)
    hw, st = int(g[f"{tag}_hw"][0]), int(g[f"stride_{tag}"][0])
    img = synthetic.tile_pixels(int(g[f"{tag}_img_seed"][0]), hw, hw)
    e.set_image(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    res = {}
    for i, (x, ws) in enumerate(e.stage_out):
        G, C = e.grid[i], og.embed_dims[i]
        perm = torch.from_numpy(sam2.window_order(G, ws).astype(np.int64)).cuda()
        mine = x[perm][:, :C].float().cpu().flatten()[::st]
        f32, b16 = torch.from_numpy(g[f"{tag}_stage{i}_f32"]), bits_to_f32(g[f"{tag}_stage{i}_bf16"])
        (mx, rms), (rmx, rrms) = _err(mine, f32), _err(b16, f32)
        res[f"stage{i}"] = {"dev_vs_f32": [mx, rms], "hf_bf16_vs_f32": [rmx, rrms]}
        assert rms <= BAND * rrms + 1e-4 and mx <= 2.5 * rmx + 1e-3, (i, res[f"stage{i}"])
    for i, (x, C) in enumerate(((e.f0, og.fpn_dim // 8), (e.f1, og.fpn_dim // 4), (e.emb, og.fpn_dim))):
        mine = x[:, :C].float().cpu().flatten()[::st]
        f32, b16 = torch.from_numpy(g[f"{tag}_feat{i}_f32"]), bits_to_f32(g[f"{tag}_feat{i}_bf16"])
        (mx, rms), (rmx, rrms) = _err(mine, f32), _err(b16, f32)
        res[f"feat{i}"] = {"dev_vs_f32": [mx, rms], "hf_bf16_vs_f32": [rmx, rrms]}
        assert rms <= BAND * rrms + 1e-4 and mx <= 2.5 * rmx + 1e-3, (i, res[f"feat{i}"])
    for p in range(3):
        box = g[f"{tag}_p{p}_box"].tolist() or None
        pts = g[f"{tag}_p{p}_pts"]
        logits, scores, low = e.predict(pts if len(pts) else None, g[f"{tag}_p{p}_labels"] if len(pts) else None, box, return_logits=True)
        ref_low, ref_iou = g[f"{tag}_p{p}_low"], g[f"{tag}_p{p}_iou"]
        (mx, rms), (rmx, rrms) = _err(low, ref_low), _err(bits_to_f32(g[f"{tag}_p{p}_low_bf16"]), ref_low)
        res[f"p{p}"] = {"low_dev_vs_f32": [mx, rms], "low_hf_bf16_vs_f32": [rmx, rrms], "iou_dev": scores.tolist(), "iou_f32": ref_iou.tolist()}
        assert rms <= BAND * rrms and mx <= 2.5 * rmx, (p, res[f"p{p}"])
        assert np.abs(scores - ref_iou).max() <= max(3 * np.abs(bits_to_f32(g[f"{tag}_p{p}_iou_bf16"]).numpy() - ref_iou).max(), 0.01), (scores, ref_iou)
        best = int(np.argmax(ref_iou))
        assert int(np.argmax(scores)) == best
        # the selected mask: exact wherever the float32 logit clears the bf16 band, and the object union kernel agrees with predict()
        _, _, up = S.postprocess(torch.from_numpy(ref_low), torch.from_numpy(ref_iou), (hw, hw))
        clear = up[best].abs() > 2.5 * rmx
        mine = torch.from_numpy(logits[best] > 0)
        want = torch.from_numpy(np.unpackbits(g[f"{tag}_p{p}_mask_bits"])[: hw * hw].reshape(hw, hw).astype(bool))
        assert bool((mine == want)[clear].all()), p
        res[f"p{p}"]["mask_pixels_differing"] = int((mine != want).sum())
        res[f"p{p}"]["pixels_inside_band"] = int((~clear).sum())
        assert (mine != want).sum() <= (~clear).sum()
        acc = torch.zeros(hw, hw, dtype=torch.uint8, device="cuda")
        e.predict_or(acc, pts if len(pts) else None, g[f"{tag}_p{p}_labels"] if len(pts) else None, box)
        assert torch.equal(acc.cpu().bool(), mine)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "sam2_parity_r03.json")
    cur = json.load(open(path)) if os.path.exists(path) else {}
    cur[tag] = res
    json.dump(cur, open(path, "w"), indent=1)
    print(tag, json.dumps(res))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pipeline_two_stage_flow_with_sam2_on_device(golden_dir, tmp_path, dtype):
    """The reference's run() sequence with the REAL predictor behind seg_infer (SAM2 on the device, tiny geometry, the oracle's synthetic
    weights) instead of the synthetic stand-in: stage-1 / stage-2 PNGs equal an independent replay (second engine, per-object predict ->
    arg-max -> OR -> nearest 756 -> 768 with the oracle's raster ops).  float32 (the reference's precision, seg_infer's default): the
    PNGs ARE the float32 oracle's masks (a pixel whose logit is within float32 round-off of 0 may differ: at most 8 of 4.7 M);
    bf16: within the bf16 band of them."""
    import json
    from PIL import Image
    from oracle import host_ref as H
    from oracle import sam2_ref as S
    from roll.pipeline.rlvr import rlvr_socioseg_vlm_pipeline_infer as P
    from socioreasoner_amd import sam2, socioseg_data
    from socioreasoner_amd.config import geometry_tiny
    from socioreasoner_amd.textproc import SyntheticProcessor
    from tests.test_gpu_pipeline import _canned_boxes, _cfg, _scripted_worker
    geom = geometry_tiny()
    proc = SyntheticProcessor(geom)
    cfg = _cfg(tmp_path, resp=400, prompt=2200)
    cfg.actor_infer.generating_args["temperature"] = 0
    seen = {"stage2_images": {}, "stage2_text": {}}
    w = _scripted_worker(cfg, geom, proc, seen)
    e1, og = _engine("tiny", dtype)
    samples = socioseg_data.synthetic_socioseg(4)
    served = sam2.Sam2Predictor(e1)
    pipe = P.SocioSegInferPipeline(cfg, dataset=samples, processor=proc, actor_worker=w, sam_predictor_provider=lambda **_: served)
    acc = pipe.run()
    # seg_infer went through segment_batch: the encoder ran once per satellite image (stage 2 found stage 1's embedding in the cache)
    assert served.stats["encoded"] <= len(samples) and served.stats["cache_hits"] >= 1, served.stats
    assert served.stats["encoder_passes"] < served.stats["images"], served.stats
    res = os.path.join(str(tmp_path), "result")
    e2, _ = _engine("tiny", dtype)
    replay = sam2.Sam2Predictor(e2)
    oracle = S.Sam2Oracle(S.synthetic_weights(og), og)
    n_diff = n_px = 0
    ious = []
    for s in samples:
        q = s["problem"]
        seg = socioseg_data.load_image(s["sat_image"]).convert("RGB").resize((756, 756))
        replay.set_image(seg)
        oracle.set_image(np.asarray(seg))

        def masks_for(prompts, pred):
            acc_m = np.zeros((756, 756), np.uint8)
            for pr in prompts:
                m, sc, _ = pred.predict(**pr)
                acc_m = H.mask_union([acc_m, np.asarray(m[int(np.argmax(sc))]).astype(np.uint8)])
            return H.resize_nearest(acc_m, 768, 768)
        boxes1 = [] if q == "school" else _canned_boxes(q)
        pr1 = [{"box": np.array(b)} for b in boxes1]
        pr2 = [{"box": np.array(b), "point_coords": np.array([[(b[0] + b[2]) // 2, (b[1] + b[3]) // 2], [b[2] + 15, b[3] + 15]]), "point_labels": np.array([1, 1])}
               for b in boxes1]
        for stage, prompts in (("stage1", pr1), ("stage2", pr2)):
            got = np.asarray(Image.open(os.path.join(res, stage, s["id"] + ".png")))
            assert np.array_equal(got, masks_for(prompts, replay) * 255), (s["id"], stage)
            want32 = masks_for(prompts, oracle)
            n_diff += int((got != want32 * 255).sum())
            n_px += got.size
            if stage == "stage2":
                ious.append(H.compute_giou(got // 255, np.asarray(s["mask_label"].convert("L"))))
    assert abs(acc - float(np.mean(ious))) < 1e-12
    if dtype == torch.float32:
        assert n_diff <= 8, (n_diff, n_px)                # the reference's precision: the oracle's masks themselves
    else:
        assert n_diff <= 0.02 * n_px, (n_diff, n_px)      # bf16 device masks vs the float32 oracle's: boundary pixels only
    print("sam2 pipeline: pixels differing from the float32 oracle", n_diff, "of", n_px, "giou_acc", acc)


@pytest.mark.parametrize("tag", ["tiny", "large"])
def test_sam2_batched_encoder_equals_one_image_at_a_time_and_cache_replays(tag):
    """set_images over B images stacks their tokens in every launch; the features of image b must be the ones a set_image of that image
    alone leaves (same kernels, same order of arithmetic per row => bit-for-bit), and the predictor's embedding cache must hand them back."""
    from socioreasoner_amd import sam2, synthetic
    e, og = _engine(tag)
    hw = 189 if tag == "tiny" else 756
    sc = hw / 189.0
    imgs = [torch.from_numpy(synthetic.tile_pixels(40 + i, hw, hw)).cuda() for i in range(3)]
    single = []
    for im in imgs:
        e.set_image(im)
        single.append(e.features())
    e.set_images(imgs)
    for b in range(3):
        e.select(b)
        ft = e.features()
        for k in ("emb", "keys0", "f0", "f1"):
            assert torch.equal(ft[k].view(torch.int16), single[b][k].view(torch.int16)), (b, k)
    pr = sam2.Sam2Predictor(e, batch=2)
    host = [im.cpu().numpy() for im in imgs]
    prompts = [[{"point_coords": [[(60 + 9 * i) * sc, 70 * sc]], "point_labels": [1], "box": [20 * sc, 30 * sc, 150 * sc, 160 * sc]},
                {"box": [10 * sc, 10 * sc, 90 * sc, 120 * sc]}, {"oops": 1}] for i in range(3)]
    a = pr.segment_batch(host, prompts)
    assert pr.stats["encoded"] == 3 and pr.stats["encoder_passes"] == 2 and pr.stats["cache_hits"] == 0
    b_ = pr.segment_batch(host[::-1] + [host[0]], prompts[::-1] + [prompts[0]])       # stage 2: same images again (and one twice)
    assert pr.stats["encoded"] == 3 and pr.stats["cache_hits"] == 3
    for x, y in zip(a + [a[0]], b_[2::-1] + [b_[3]]):
        assert torch.equal(x, y)
    for i in range(3):                                                                  # against the one-image path
        e.set_image(imgs[i])
        assert torch.equal(pr.segment_objects(prompts[i]), a[i])
        assert int(a[i].sum()) > 0


@pytest.mark.parametrize("tag", ["tiny", "large"])
def test_sam2_objects_decoded_together_equal_one_at_a_time(tag):
    """or_objects stacks the objects of an image along the rows of every decoder launch (and replays a captured launch sequence): logits,
    scores and the union must be what the one-object calls give, for mixed prompt lengths and more objects than one pass takes."""
    from socioreasoner_amd import synthetic
    e, og = _engine(tag)
    hw = 189 if tag == "tiny" else 756
    sc = hw / 189.0
    e.set_image(torch.from_numpy(synthetic.tile_pixels(77, hw, hw)).cuda())
    prompts = [dict(box=[(20 + 3 * i) * sc, 30 * sc, (120 + 4 * i) * sc, 150 * sc]) if i % 3 == 0 else
               dict(box=[10 * sc, (15 + 2 * i) * sc, (90 + i) * sc, 140 * sc], point_coords=[[(50 + i) * sc, (60 + k) * sc] for k in range(1 + i % 2)],
                    point_labels=[1] * (1 + i % 2)) for i in range(11)]
    prompts.append(dict(point_coords=[[(30 + 5 * k) * sc, (40 + 3 * k) * sc] for k in range(12)], point_labels=[1, 0] * 6))      # 19 tokens: the wide path
    m = e.grid[0]
    singles, acc1 = [], torch.zeros(hw, hw, dtype=torch.uint8, device="cuda")
    for p in prompts:
        low, iou = e.predict_or(acc1, **p)
        singles.append((low.clone(), iou.clone()))
    for graph in (True, False):
        e.graph_decode = graph
        for _ in range(2):                      # second round: replayed graphs
            acc2 = torch.zeros_like(acc1)
            e.predict_or_many(acc2, prompts)
            assert torch.equal(acc1, acc2) and int(acc1.sum()) > 0
    e.graph_decode = True
    prep = sorted((e.prompt(p.get("point_coords"), p.get("point_labels"), p.get("box")) for p in prompts[:5]), key=lambda cl: len(cl[1]))
    order = sorted(range(5), key=lambda i: len(e.prompt(prompts[i].get("point_coords"), prompts[i].get("point_labels"), prompts[i].get("box"))[1]))
    low, iou = e.decode_many(prep)
    for r, i in enumerate(order):
        assert torch.equal(low[r * m * m:(r + 1) * m * m], singles[i][0]) and torch.equal(iou[r], singles[i][1][0]), (r, i)
