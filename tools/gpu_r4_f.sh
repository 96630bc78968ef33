#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r04_f_tests.log
tail -8 gpurun_out/r04_f_tests.log
for b in 64 128; do
timeout 300 python bench.py --batch $b --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('static $b', d['value'], d['phase_ms_per_step'], d['roofline']['avg_launch_us'])"
done
timeout 600 python bench.py --batch 128 --steps 2 --warmup 1 --waves 2 --no-latency --no-cpu-baseline --no-sam 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cont 128', d['value'], d['roofline']['decode_step_ms'], d['phase_ms_per_step']['scheduler'])"
