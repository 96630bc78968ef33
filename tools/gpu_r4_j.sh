#!/bin/bash
cd "$(dirname "$0")/.."
for g in 2 1; do for b in 64 128; do
SR_NARROW_G=$g timeout 300 python bench.py --batch $b --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('narrowG=$g static $b', d['value'], 'decode', d['phase_ms_per_step']['decode'], 'checksum', d['result_checksum'])"
done; done
