#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 python tools/probe_abort.py 2>&1 | grep -v Warn | tail -8
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -k "abort or above_32" 2>&1 | tail -12
timeout 1200 python bench.py --no-cpu-baseline > gpurun_out/r04_e_bench.json 2> gpurun_out/r04_e_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_e_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms_per_step',d['ms_per_step'],'drained',d.get('drained_step'))
print('more_rows',json.dumps(d.get('more_rows_per_gpu')))
print('sam2 f32', d['sam2']['float32']['tiles_per_s_4_objects_batched'], 'bf16', d['sam2']['bf16']['tiles_per_s_4_objects_batched'])
PY
tail -3 gpurun_out/r04_e_bench.err
