#!/usr/bin/env python3
"""Benchmark of the SocioReasoner-3B inference hot path on MI355X (contract: see the round prompt / DESIGN.md).

One "step" = one pass of the hot path over one batch of synthetic tiles per GPU:
  uint8 448x448 tile -> patchify -> ViT (1024 patches) -> merger (256 tokens) -> LM prefill (448-token prompt)
  -> greedy decode of exactly 128 tokens (EOS ignored) -> raster tail (union of 4 756^2 masks -> nearest 768^2 -> IoU).
Default workload = BASELINE.json configs[1] (SocioReasoner-3B bf16, 1 x MI355X, batch 1, greedy decode);
``--batch 32`` runs configs[2]'s batch size.  Inputs (tiles, masks, weights) are resident in HBM before timing.
Multi-GPU: one process per GPU (torchrun), tiles sharded data-parallel, no data-path collective while generating,
one RCCL all-gather of the per-tile results (tokens + IoU counts) per step -> weak scaling.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16
# algorithmic work per tile
VIT_GFLOP, PREFILL_GFLOP = 1342.9, 2516.3
N_NEW = 128


def lm_weight_bytes(g):
    t = g.text
    qn = (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
    per_layer = (qn * t.hidden_size + t.hidden_size * t.num_attention_heads * t.head_dim + 3 * t.intermediate_size * t.hidden_size) * 2
    return per_layer * t.num_hidden_layers, t.vocab_size * t.hidden_size * 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="tiles per GPU per step (1 = configs[1], 32 = configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--fp8", action="store_true", help="BASELINE.json configs[4] weights: LM decoder linears fp8 e4m3, per-channel scale")
    ap.add_argument("--continuous", action="store_true",
                    help="serve the tiles through the continuous-batching scheduler (configs[2]: admit on finish, 2x batch requests)")
    ap.add_argument("--gather-logits", action="store_true",
                    help="verification mode: all-gather the float32 logits of every decode step (north_star's literal exchange)")
    args = ap.parse_args()

    from socioreasoner_amd import dp, hostops, raster, synthetic
    from socioreasoner_amd.config import geometry_3b
    from socioreasoner_amd.engine import Engine
    from socioreasoner_amd import lib as L

    rank, world, local = dp.init_distributed()
    assert world == args.gpus or world == 1, (world, args.gpus)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    B = args.batch
    geom = geometry_3b()
    eng = Engine(geom, max_patches=1024 * B, max_prefill_tokens=448 * B, max_batch=B, max_ctx=640, max_new_tokens=N_NEW, device=str(dev), lm_fp8=args.fp8)
    t0 = time.time()
    eng.load_synthetic_weights(seed=0)
    load_s = time.time() - t0

    # ---- synthetic inputs, resident in HBM
    grid = (1, 32, 32)
    tiles = [rank * B + i for i in range(B)]
    imgs = [torch.from_numpy(synthetic.tile_pixels(i)).to(dev) for i in tiles]
    ids = [synthetic.tile_prompt(geom, i, grid) for i in tiles]
    pos3 = []
    for x in ids:
        p, _ = hostops.get_rope_index(torch.from_numpy(x)[None], [grid], None, image_token_id=geom.image_token_id,
                                      vision_start_token_id=geom.vision_start_token_id)
        pos3.append(p[:, 0].numpy())
    masks, gts = [], []
    for i in tiles:
        m, g = synthetic.tile_masks(i)
        masks.append(torch.from_numpy(m).to(dev))
        gts.append(torch.from_numpy(g).to(dev))
    grids = [grid] * B
    ev = lambda: torch.cuda.Event(enable_timing=True)
    phase_ms = {"vit": 0.0, "prefill": 0.0, "decode": 0.0, "raster": 0.0}

    def step_continuous():
        """2*B requests through B rows: the second half is admitted as rows free up (EOS is ignored by the metric, so all
        rows of a wave finish together; the point is the measured cost of the request-level path)."""
        from socioreasoner_amd.serving import ContinuousBatcher, Request
        cb = ContinuousBatcher(eng, eos=[], pad_id=0, steps_per_poll=16)
        reqs = [Request(ids=ids[k % B], pos3=pos3[k % B], max_new=N_NEW, images=[imgs[k % B]], grids=[grid]) for k in range(2 * B)]
        toks = cb.run(reqs)
        counts = torch.empty(2 * B, 2, dtype=torch.int64, device=dev)
        for k in range(2 * B):
            acc = torch.zeros(756, 756, dtype=torch.uint8, device=dev)
            for m in masks[k % B]:
                raster.mask_union_(acc, m)
            counts[k] = raster.iou_counts(raster.resize_nearest(acc, 768, 768), gts[k % B])
        res = torch.cat([torch.tensor(toks, dtype=torch.int64, device=dev), counts], dim=1)
        if world > 1:
            res = dp.all_gather_rows(res, 2 * B * world)
        return res

    def step(record=False):
        if args.continuous:
            return step_continuous()
        e0, e1, e2, e3, e4 = ev(), ev(), ev(), ev(), ev()
        e0.record()
        pix = torch.cat([eng.patchify(im) for im in imgs], dim=0)
        emb = eng.vit_forward(pix, grids)
        e1.record()
        first = eng.prefill(ids, pos3, emb, return_logits=args.gather_logits)
        e2.record()
        if args.gather_logits:
            alltoks, bad = dp.decode_with_logits_gather(lambda t: eng.decode_step(t), first, N_NEW, B * world)
            assert bad == 0, f"{bad} on-device argmax results differ from the argmax of the gathered logits"
            toks = alltoks[rank * B:(rank + 1) * B]
        else:
            toks = eng.decode(N_NEW, use_graph=not args.no_graph)
        e3.record()
        counts = torch.empty(B, 2, dtype=torch.int64, device=dev)
        for b in range(B):
            acc = torch.zeros(756, 756, dtype=torch.uint8, device=dev)
            for m in masks[b]:
                raster.mask_union_(acc, m)
            counts[b] = raster.iou_counts(raster.resize_nearest(acc, 768, 768), gts[b])
        e4.record()
        res = torch.cat([toks.to(torch.int64), counts], dim=1)             # [B, 130] per-tile result row
        if world > 1:
            res = dp.all_gather_rows(res, B * world)                       # the one RCCL exchange of the step
        if record:
            torch.cuda.synchronize(dev)
            for k, a, b_ in (("vit", e0, e1), ("prefill", e1, e2), ("decode", e2, e3), ("raster", e3, e4)):
                phase_ms[k] += a.elapsed_time(b_)
        return res

    if world > 1:      # open the RCCL communicator outside the timed region even with --warmup 0
        dp.all_gather_rows(torch.zeros(B, 1, dtype=torch.int64, device=dev), B * world)
    for _ in range(args.warmup):
        step()
    dp.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(record=True)
    torch.cuda.synchronize(dev)
    dp.barrier()
    dt = time.perf_counter() - t0
    dt = dp.all_reduce_max(dt, dev)
    tiles_per_s = world * B * (2 if args.continuous else 1) * args.steps / dt

    # ---- roofline of the dominant kernel (k_gemv, the decode weight stream): HIP events on the launch stream around
    # the exact per-step launch sequence of that kernel (145 launches: 4 per layer + LM head) on weight-sized operands
    out = {}
    if rank == 0:
        lib = L.load()
        t_ = geom.text
        H, QN, I = t_.hidden_size, (t_.num_attention_heads + 2 * t_.num_key_value_heads) * t_.head_dim, t_.intermediate_size
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        P = lambda x: C.c_void_p(x.data_ptr())
        nl = t_.num_hidden_layers
        # distinct weight copies per layer so that nothing is served from L2 / Infinity Cache (256 MB) between launches
        wq = torch.empty(nl, QN, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
        wo = torch.empty(nl, H, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
        wg = torch.empty(nl, 2 * I, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
        wd = torch.empty(nl, H, I, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
        wv = torch.empty(t_.vocab_size, H, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
        x = torch.empty(B, I, dtype=torch.bfloat16, device=dev).normal_(0, 1)
        part = torch.empty(4, B, QN, dtype=torch.float32, device=dev)
        act = torch.empty(B, I, dtype=torch.bfloat16, device=dev)
        lg = torch.empty(B, t_.vocab_size, dtype=torch.float32, device=dev)

        nw = torch.ones(H, dtype=torch.bfloat16, device=dev)
        bq = torch.zeros(QN, dtype=torch.bfloat16, device=dev)
        slabs = torch.zeros(2, B, H, dtype=torch.float32, device=dev)
        xo = torch.zeros(B, H, dtype=torch.bfloat16, device=dev)
        qkv_o = torch.empty(B, QN, dtype=torch.bfloat16, device=dev)
        xr = torch.zeros(B, H, dtype=torch.bfloat16, device=dev)
        nb = lib.sr_op_gemv_f32_blocks(t_.vocab_size, B, H, 1 if B <= 4 else 0)
        av = torch.empty(B, nb, dtype=torch.float32, device=dev)
        ai = torch.empty(B, nb, dtype=torch.int32, device=dev)
        fused = B <= 4          # same launch configuration as the engine's decode layer (engine.hip enqueue_decode_forward)
        TL = 0x100              # weights are fragment-ordered in the engine; the timing does not depend on the values
        eps = C.c_float(1e-6)

        if args.fp8:            # fp8 images + scales of the four layer linears (values do not matter for the timing)
            w8 = [torch.empty(nl, n_ * k_, dtype=torch.uint8, device=dev).random_(0, 120) for n_, k_ in ((QN, H), (H, H), (2 * I, H), (H, I))]
            sc8 = torch.ones(2 * I, dtype=torch.float32, device=dev)

        def gemv_sequence():
            if args.fp8:
                for l in range(nl):
                    lib.sr_op_gemv_f8(P(x), I, P(w8[0][l]), P(sc8), B, QN, H, P(qkv_o), QN, 3, P(bq), P(nw) if fused else None, eps, 1, s)
                    lib.sr_op_gemv_f8(P(x), I, P(w8[1][l]), P(sc8), B, H, H, P(xr), H, 4, None, None, eps, 1, s)
                    lib.sr_op_gemv_f8(P(x), I, P(w8[2][l]), P(sc8), B, 2 * I, H, P(act), I, 1, None, P(nw) if fused else None, eps, 1, s)
                    lib.sr_op_gemv_f8(P(act), I, P(w8[3][l]), P(sc8), B, H, I, P(part), H, 0, None, None, eps, 2, s)
                lib.sr_op_gemv_fused(P(x), I, P(wv), B, t_.vocab_size, H, P(lg), t_.vocab_size, 2 | TL, None, P(nw) if fused else None, eps,
                                     P(slabs) if fused else None, 2 if fused else 0, P(xo) if fused else None, P(av), P(ai), s)
                return
            for l in range(nl):
                lib.sr_op_gemv_fused(P(x), I, P(wq[l]), B, QN, H, P(qkv_o), QN, 3 | TL, P(bq), P(nw) if fused else None, eps,
                                     P(slabs) if fused else None, 2 if fused else 0, P(xo) if fused else None, None, None, s)
                lib.sr_op_gemv_fused(P(x), I, P(wo[l]), B, H, H, P(xr), H, 4 | TL, None, None, eps, None, 0, None, None, None, s)
                lib.sr_op_gemv_fused(P(x), I, P(wg[l]), B, 2 * I, H, P(act), I, 1 | TL, None, P(nw) if fused else None, eps, None, 0, None, None, None, s)
                lib.sr_op_gemv(P(act), I, P(wd[l]), B, H, I, P(part), 2, 0 | TL, s)
            lib.sr_op_gemv_fused(P(x), I, P(wv), B, t_.vocab_size, H, P(lg), t_.vocab_size, 2 | TL, None, P(nw) if fused else None, eps,
                                 P(slabs) if fused else None, 2 if fused else 0, P(xo) if fused else None, P(av), P(ai), s)
        gemv_sequence()
        a, b_ = ev(), ev()
        reps = 5
        a.record()
        for _ in range(reps):
            gemv_sequence()
        b_.record()
        torch.cuda.synchronize(dev)
        n_launch = 4 * nl + 1
        avg_ms = a.elapsed_time(b_) / reps / n_launch
        wl, wh = lm_weight_bytes(geom)
        if args.fp8:
            wl = wl / 2          # the layer linears stream 1 byte per weight (+ 4 bytes per output channel, < 0.1 %)
        bytes_per_launch = (wl + wh) / n_launch
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        decode_step_ms = phase_ms["decode"] / args.steps / (N_NEW - 1)
        kv_bytes = 36864.0 * (448 + N_NEW / 2) * B
        # HBM traffic per launch from the PMC passes committed under profiles/ (FETCH_SIZE doubled per the gfx950 note):
        # measured ratio traffic / algorithmic bytes of this kernel family x the algorithmic bytes of one launch
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_gemv_traffic.json")))
            traffic = round(pmc["traffic_over_algorithmic_weighted"] * bytes_per_launch)
        except Exception:  # noqa: BLE001
            pass
        if args.fp8:
            traffic = None       # the PMC calibration under profiles/ was taken on the bf16 stream
        roof = {"bound": "hbm", "kernel": "k_gemv (decode weight stream, all LM linears + LM head)" + (" [fp8 layer linears]" if args.fp8 else ""),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "bytes_per_launch": round(bytes_per_launch), "avg_launch_us": round(avg_ms * 1e3, 2),
                "launches_per_decode_step": n_launch,
                "decode_step_ms": round(decode_step_ms, 4) if decode_step_ms > 0 else None,
                "decode_step_achieved_GBs": round((wl + wh + kv_bytes) / (decode_step_ms * 1e-3) / 1e9, 1) if decode_step_ms > 0 else None}
        del wq, wo, wg, wd, wv
        vit_ms = phase_ms["vit"] / args.steps
        pre_ms = phase_ms["prefill"] / args.steps
        phases = {k: round(v / args.steps, 3) for k, v in phase_ms.items()}
        if not args.continuous:      # the request-level path interleaves the phases: no per-phase split
            phases["vit_mfma_frac"] = round(VIT_GFLOP * B / (vit_ms * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4)
            phases["prefill_mfma_frac"] = round(PREFILL_GFLOP * B / (pre_ms * 1e-3) / 1e3 / MFMA_PEAK_TFLOPS, 4)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline()
        out = {
            "metric": "satellite tiles/sec (448x448, SocioReasoner-3B)", "value": round(tiles_per_s, 4), "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if not args.fp8 else "bf16 (fp8-e4m3 LM linear weights, bf16 activations / MFMA)", "data": "synthetic",
            "config": {"workload": f"SocioReasoner-3B {'fp8-weight' if args.fp8 else 'bf16'}, batch={B} tile(s)/GPU, 448x448 synthetic tiles, 448-token prompt, "
                                   f"greedy decode of {N_NEW} tokens (EOS ignored), raster tail; random-init weights "
                                   f"(counter-based generator, seed 0)" + (" [BASELINE.json configs[1]]" if B == 1 and not args.fp8 else ""),
                       "tiles_per_gpu_per_step": B * (2 if args.continuous else 1),
                       "scheduling": "continuous batching (admit on finish) through B rows" if args.continuous else "static batch",
                       "parallelism": f"dp{world}", "decode": "hipGraph" if not args.no_graph else "eager",
                       "exchange": "float32 logits all-gather per decode step (verification mode)" if args.gather_logits
                       else "one all-gather of 1 KB result rows per tile"},
            "roofline": roof, "cpu_baseline": cpu, "phase_ms_per_step": phases,
            "weights_load_s": round(load_s, 1), "workspace_GB": round(eng.workspace_bytes / 1e9, 2),
            "result_checksum": int(res.sum().item()),
        }
        print(json.dumps(out), flush=True)
    dp.barrier()
    eng.close()


def cpu_baseline():
    """The oracle (a port of the reference's HF-eager CPU path) on this host's cores, bounded sample:
    2 of 32 ViT blocks and 2 of 36 LM layers at the true dimensions on the same synthetic tile (448-token prefill,
    3 decode steps), plus patch-embed / merger / LM head once; extrapolated linearly in depth and decode steps."""
    from oracle import host_ref as H
    from oracle import model_ref as MR
    from oracle import weights as WG
    from socioreasoner_amd import hostops, synthetic
    cfg = MR.config_3b()
    cfg.vision.depth, cfg.text.num_hidden_layers = 4, 4          # the sample: 4 ViT blocks, 4 LM layers at the true dimensions
    cfg.vision.fullatt_block_indexes = (1, 3)
    W = WG.LazyWeights(cfg, seed=0)
    for n, _, _ in WG.param_specs(cfg):
        W[n]                                  # materialise outside the timed region
    img = synthetic.tile_pixels(0)
    grid = (1, 32, 32)
    from socioreasoner_amd.config import geometry_3b
    ids = synthetic.tile_prompt(geometry_3b(), 0, grid)
    p3, _ = hostops.get_rope_index(torch.from_numpy(ids)[None], [grid], None)
    p3 = p3[:, 0]
    t0 = time.perf_counter()
    pv, _ = H.patchify(img)
    vc = cfg.vision
    widx, cu_win = MR.vision_window_index([grid], 2, 112, 14)
    cu_full = MR.vision_full_seqlens([grid])
    x = MR.linear(MR.r(torch.from_numpy(pv)), W["visual.patch_embed.proj.weight"])
    x = x.reshape(256, 4, -1)[widx].reshape(1024, -1)
    cos, sin = MR.vit_rotary_tables(vc, [grid], widx)
    t1 = time.perf_counter()
    xw = MR.vit_block(W, 0, vc, x, cu_win, cos, sin)
    xw = MR.vit_block(W, 2, vc, xw, cu_win, cos, sin)
    t2 = time.perf_counter()
    xf = MR.vit_block(W, 1, vc, xw, cu_full, cos, sin)
    xf = MR.vit_block(W, 3, vc, xf, cu_full, cos, sin)
    t3 = time.perf_counter()
    emb = MR.vit_merger(W, vc, xf)[torch.argsort(widx)]
    t4 = time.perf_counter()
    vit_s = (t1 - t0) + 28 * (t2 - t1) / 2 + 4 * (t3 - t2) / 2 + (t4 - t3)
    tc = cfg.text
    caches = MR.new_caches(cfg)
    NL = 4                                      # LM layers in the sample
    t5 = time.perf_counter()
    h = MR.embed_with_images(W, cfg, torch.from_numpy(ids), emb)
    c_, s_ = MR.mrope_tables(tc, p3)
    t6 = time.perf_counter()
    for i in range(NL):
        h = MR.lm_layer(W, i, tc, h, c_, s_, caches[i])
    t7 = time.perf_counter()
    lg = MR.rmsnorm(h[-1:], W["model.norm.weight"], tc.rms_norm_eps) @ W["lm_head.weight"].t()
    t8 = time.perf_counter()
    prefill_s = 36 / NL * (t7 - t6) + (t8 - t7)
    nd = 12
    t9 = time.perf_counter()
    for k in range(nd):
        xx = W["model.embed_tokens.weight"][torch.tensor([int(lg.argmax())])]
        cc, ss = MR.mrope_tables(tc, torch.full((3, 1), int(p3.max()) + 1 + k))
        for i in range(NL):
            xx = MR.lm_layer(W, i, tc, xx, cc, ss, caches[i])
    t10 = time.perf_counter()
    head_s = t8 - t7
    decode_s = (N_NEW - 1) * (36 / NL * (t10 - t9) / nd + head_s)
    total = vit_s + prefill_s + decode_s
    return {"value": round(1.0 / total, 5), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle/model_ref.py (float32 math, HF bf16 rounding points) on one synthetic 448x448 tile: 4/32 ViT "
                      "blocks (2 window + 2 full) and 4/36 LM layers at true dims, 448-token prefill, 12 decode steps, "
                      "extrapolated linearly in depth and to 127 decode steps",
            "seconds_per_tile_extrapolated": round(total, 2),
            "phases_s": {"vit": round(vit_s, 2), "prefill": round(prefill_s, 2), "decode": round(decode_s, 2)}}


if __name__ == "__main__":
    main()
