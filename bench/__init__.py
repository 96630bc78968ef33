"""The measurement contract of this repository, split into auditable parts:

    bench.py            entry point: arguments, rank launch, the timed region, ONE JSON line
    bench/common.py     constants (peaks, algorithmic work per tile), the argument parser, small helpers
    bench/launch.py     `--gpus N` outside torchrun: start the N ranks
    bench/workload.py   the synthetic workload: inputs resident in HBM, one step (static batch / continuous batching), the raster tail
    bench/roofline.py   the `roofline` object: decode weight stream timed IN SITU (rocprofv3 kernel trace of a short static decode, child process),
                        the launch-only replay beside it, HBM traffic from PMC passes, decode_step_frac
    bench/side.py       side measurements carried in the same line (static batch, drained step, 64 / 128 rows, ragged lengths, SAM2, pipeline, batch 1)
    bench/baselines.py  `cpu_baseline`: the oracle / HF-bf16 on the host cores (the only place outside tests/ and smoke() that touches oracle/)
    bench/line.py       assembly of the JSON line
"""
