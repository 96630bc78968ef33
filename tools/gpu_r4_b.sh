#!/bin/bash
# round 4, second GPU call: the whole GPU suite, the rows-step probe, the default bench line (steps served as one stream)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 > gpurun_out/r04_b_tests.log
tail -15 gpurun_out/r04_b_tests.log
timeout 600 python tools/probe_rows_step.py > gpurun_out/r04_b_probe_rows.log 2>&1
cat gpurun_out/r04_b_probe_rows.log | grep -v Warning | tail -30
timeout 900 python bench.py --no-cpu-baseline --no-sam > gpurun_out/r04_b_bench.json 2> gpurun_out/r04_b_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_b_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms_per_step',d['ms_per_step'],'drained',d.get('drained_step'),'static',d['static_batch']['decode_step_ms'],'roof',d['roofline']['frac'],d['roofline']['decode_step_ms'])
print(json.dumps(d['phase_ms_per_step'])[:900])
PY
tail -3 gpurun_out/r04_b_bench.err
