#!/bin/bash
# the other bench configurations on the round's final code (no test suite: tools/gpu_r4_suite.sh runs that)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { timeout 600 python bench.py "$@" --no-latency --no-cpu-baseline --no-sam --no-more-rows 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', d['value'], d['unit'], 'ms/step', d['ms_per_step'])"; }
run --pair --steps 2 --warmup 1
run --fp8-mx --tile 896 --batch 16 --static --steps 2 --warmup 1
run --fp8 --steps 2 --warmup 1
run --no-overlap --steps 2 --warmup 1
run --batch 64 --steps 2 --warmup 1 --waves 4
run --batch 128 --steps 2 --warmup 1 --waves 4
