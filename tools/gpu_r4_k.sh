#!/bin/bash
# full suite, then the other bench configurations (sanity of every mode after round 4's changes)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r04_k_tests.log
tail -8 gpurun_out/r04_k_tests.log
run() { timeout 600 python bench.py "$@" --no-latency --no-cpu-baseline --no-sam 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', d['value'], d['unit'], 'ms/step', d['ms_per_step'])"; }
run --pair --steps 2 --warmup 1
run --fp8-mx --tile 896 --batch 16 --static --steps 2 --warmup 1
run --fp8 --steps 2 --warmup 1
run --batch 1 --steps 3 --warmup 1
run --drain --steps 2 --warmup 1
run --no-overlap --steps 2 --warmup 1
run --batch 64 --steps 2 --warmup 1 --waves 4
run --batch 128 --steps 2 --warmup 1 --waves 4
