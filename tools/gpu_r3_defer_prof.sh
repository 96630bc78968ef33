#!/bin/bash
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out
VARIANTS="0 1 0 1" TESTK="truth" bash tools/gpu_r3_defer.sh
cd /tmp; export TMPDIR=/tmp
SR_DEFER_LN=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/dprof -o d1 -- python $R/bench.py --static --steps 1 --warmup 1 --no-latency --no-cpu-baseline --no-sam > $R/gpurun_out/defer_prof.log 2>&1
cd $R
python tools/rocpd_stats.py $(find /tmp/dprof -name 'd1_results.db' | head -1) gpurun_out/r03_defer1_kernel_stats.md > /dev/null
head -16 gpurun_out/r03_defer1_kernel_stats.md | cut -c1-150
