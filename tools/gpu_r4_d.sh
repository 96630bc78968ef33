#!/bin/bash
# round 4, fourth GPU call: more than 32 batch rows -- op tests, tiny rows-mode test, full-depth 64 / 128-row decode vs HF, then the 64- and 128-row bench lines
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round4.py -q -m gpu -k "fragment_ordered or above_32 or batch32_decode_vs_hf or continuous_batching_32" 2>&1 | tail -40 > gpurun_out/r04_d_tests.log
tail -25 gpurun_out/r04_d_tests.log
for b in 64 128; do
  timeout 600 python bench.py --batch $b --steps 1 --warmup 1 --waves 2 --no-latency --no-cpu-baseline --no-sam > gpurun_out/r04_d_bench_b$b.json 2> gpurun_out/r04_d_bench_b$b.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r04_d_bench_b$b.json').read().strip().splitlines()[-1])
    print('batch $b value',d['value'],'ms_per_step',d['ms_per_step'],'roof',d['roofline']['frac'],d['roofline']['avg_launch_us'],'decode_step',d['roofline']['decode_step_ms'])
    print(json.dumps(d['phase_ms_per_step'])[:700])
except Exception as e:
    print('batch $b failed', e); print(open('gpurun_out/r04_d_bench_b$b.err').read()[-1500:])
PY
  timeout 300 python bench.py --batch $b --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('static $b', d['value'], d['phase_ms_per_step'])"
done
