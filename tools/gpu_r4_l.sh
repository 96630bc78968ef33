#!/bin/bash
cd "$(dirname "$0")/.."
timeout 300 python bench.py --batch 128 --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('static 128', d['value'], 'decode', d['phase_ms_per_step']['decode'], 'checksum', d['result_checksum'])"
timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu -k "batch32_decode_vs_hf" 2>&1 | tail -3
