#!/bin/bash
# round 4, third GPU call: suite after the library clean-up, co-residency probe (128- vs 256-tile admission GEMMs next to decode, no CU masks)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r04_c_tests.log
tail -6 gpurun_out/r04_c_tests.log
SR_GEMM256=0 timeout 600 python tools/probe_coresident.py 2>&1 | grep '^{' > gpurun_out/r04_c_coresident.jsonl
timeout 600 python tools/probe_coresident.py 2>&1 | grep '^{' >> gpurun_out/r04_c_coresident.jsonl
cat gpurun_out/r04_c_coresident.jsonl
