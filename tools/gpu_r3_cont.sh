#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for flags in "" "--no-overlap"; do
  timeout 500 python bench.py --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam $flags > gpurun_out/cont.log 2>&1
  echo "flags=[$flags] exit $?"
  python - <<'PY'
import json
for l in open('gpurun_out/cont.log'):
    if l.startswith('{'):
        d = json.loads(l); p = d['phase_ms_per_step']; s = p.get('scheduler', {})
        print('value', d['value'], 'ms_per_step', d['ms_per_step'])
        print({k: v for k, v in p.items() if k != 'scheduler'})
        print(s)
        st = s.get('steps', 0) - s.get('steps_shared', 0)
        if st: print('decode alone ms/step', p['decode'] / (st / d['steps'] if st > 600 else st), ' (steps alone per bench step', st, ')')
PY
done
