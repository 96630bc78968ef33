#!/bin/bash
# round 4, first GPU call: the new tests (float32 SAM2 kernels and mode, configs[4] MX at 896, bench self-launch), then one bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_round4.py -q -m gpu 2>&1 | tail -60 > gpurun_out/r04_a_tests.log
tail -30 gpurun_out/r04_a_tests.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_a_bench.json 2> gpurun_out/r04_a_bench.err
tail -c 3000 gpurun_out/r04_a_bench.json
tail -5 gpurun_out/r04_a_bench.err
