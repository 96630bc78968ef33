#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python tools/probe_abort.py 2>&1 | grep -v Warn | tail -8
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -k "fragment_ordered or above_32 or batch32_decode_vs_hf or continuous_batching_32 or abort" 2>&1 | tail -6
for b in 64 128; do
timeout 300 python bench.py --batch $b --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('static $b', d['value'], d['phase_ms_per_step'])"
done
