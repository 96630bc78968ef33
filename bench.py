#!/usr/bin/env python3
"""Benchmark of the SocioReasoner-3B inference hot path on MI355X (contract: see the round prompt / DESIGN.md section 5).

One "step" = one pass of the hot path over one batch of synthetic tiles per GPU (bench/workload.py):
  uint8 448x448 tile -> patchify -> ViT (1024 patches) -> merger (256 tokens) -> LM prefill (448-token prompt)
  -> greedy decode of exactly 128 tokens (EOS ignored) -> raster tail (union of 4 756^2 masks -> nearest 768^2 -> IoU).

Default workload (the headline `value`) = BASELINE.json configs[2]: SocioReasoner-3B bf16, one MI355X, 32 rows in flight, CONTINUOUS BATCHING
(overlapped admission): every step serves 4 x 32 tile requests through 32 batch rows.  The same command also times configs[1] (batch 1) and
reports it under `latency_b1` of the same JSON line.  `--batch 1` makes configs[1] the headline; `--static` replaces the scheduler by one
static batch per step.  Inputs (tiles, masks, weights) are resident in HBM before timing.
Multi-GPU: one process per GPU (torchrun or `--gpus N` self-launch), tiles sharded data-parallel, no data-path collective while generating, one
RCCL all-gather of the per-tile results (tokens + IoU counts) per step -> weak scaling.  The parts live in bench/ (see bench/__init__.py).
"""
from __future__ import annotations

import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench import baselines, line, roofline, side  # noqa: E402
from bench.common import clean, parse_args  # noqa: E402
from bench.launch import self_launch  # noqa: E402
from bench.workload import Workload  # noqa: E402


def main():
    args = parse_args()
    rc = self_launch(args, os.path.abspath(__file__))
    if rc is not None:
        sys.exit(rc)
    from socioreasoner_amd import dp
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("SR_DIST_BACKEND", "nccl") == "nccl":
        # let RCCL describe its communicator (rank count, transport of every channel) to a per-rank file the exchange report parses
        os.environ.setdefault("NCCL_DEBUG", "INFO")
        os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH")
        os.environ["SR_RCCL_LOG"] = os.environ["NCCL_DEBUG_FILE"] = f"/tmp/sr_rccl_{os.getpid()}_rank{os.environ.get('RANK', '0')}.log"
    rank, world, local = dp.init_distributed()      # RCCL when ranks > 1 (gloo only if SR_DIST_BACKEND=gloo asks for it)
    if world > 1 and os.environ.get("OMP_NUM_THREADS", "").isdigit():
        torch.set_num_threads(max(1, int(os.environ["OMP_NUM_THREADS"])))      # the per-rank share of the host cores (self_launch)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s): launch `python bench.py --gpus N` (self-launching) or "
                         f"torchrun --nproc-per-node N bench.py --gpus N")
    wl = Workload(args, rank, world, local)
    dev, B = wl.dev, wl.B

    # ---- the timed region: W warm-up steps, then EXACTLY K steps between barrier + synchronize, MAX over ranks
    if world > 1:      # open the RCCL communicator outside the timed region even with --warmup 0
        dp.all_gather_rows(torch.zeros(B, 1, dtype=torch.int64, device=dev), B * world)
    if args.warmup:
        wl.run_steps(args.warmup)
    dp.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    res = wl.run_steps(args.steps, True)
    torch.cuda.synchronize(dev)
    dp.barrier()
    dt = dp.all_reduce_max(time.perf_counter() - t0, dev)

    # ---- side measurements (rank 0 of an N = 1 run; none of them inside the timed region) and the line
    if rank == 0:
        extras = world == 1 and not args.no_latency
        plain = B == 32 and args.tile == 448 and not args.pair and not args.fp8          # the headline configuration itself
        sd = {"static_batch": side.static_batch(wl) if wl.continuous and not args.no_latency else None,
              "drained_step": side.drained_step(wl) if extras and wl.continuous and not args.drain else None,
              "more_rows_per_gpu": side.more_rows() if extras and wl.continuous and plain and not args.no_more_rows else None,
              "ragged": side.ragged(wl) if extras and wl.continuous else None,
              "sam2": side.sam2(dev) if extras and not args.no_sam else None,
              "pipeline_two_stage_with_sam2": (side.pipeline(args.pipeline_samples, batch_order_too=args.pipeline_ab)
                                               if extras and wl.continuous and plain and not args.no_pipeline else None),
              "latency_b1": side.latency_b1(wl) if extras and B > 1 and not args.fp8 else None}
        roof = roofline.build(wl, line.decode_step_alone_ms(wl), sd["latency_b1"], measure=extras and not args.no_pmc)
        cpu = cpu_hf = None
        if world == 1 and not args.no_cpu_baseline:
            src = baselines.device_weight_source(dev)
            wl.eng.close()                   # the GPU is idle while the host cores are timed
            cpu = baselines.cpu_baseline(src)
            try:
                cpu_hf = baselines.cpu_baseline_hf_bf16()
            except Exception as e_:  # noqa: BLE001  (transformers missing / too little host memory: reported, never fatal)
                cpu_hf = {"value": None, "error": f"{type(e_).__name__}: {e_}"[:200]}
        print(json.dumps(clean(line.assemble(wl, dt, res, roof, cpu, cpu_hf, sd))), flush=True)
    dp.barrier()
    wl.eng.close()


if __name__ == "__main__":
    main()
