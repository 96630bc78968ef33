#!/bin/bash
# SAM2 batching check: parity tests then timings
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sam2.py -x -q -m gpu > gpurun_out/sam_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/sam_tests.log
timeout 300 python tools/bench_sam2.py > gpurun_out/sam_bench.log 2>&1
tail -5 gpurun_out/sam_tests.log; cat gpurun_out/sam_bench.log
