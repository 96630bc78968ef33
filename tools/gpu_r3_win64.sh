#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round3.py -q -m gpu -p no:cacheprovider -x -k "window_attention_without_lds" 2>&1 | tail -3
for v in 0 1 0 1; do
  SR_ATTN_WIN64=$v timeout 400 python bench.py --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam > gpurun_out/win64_$v.log 2>&1
  echo "win64=$v exit $? $(grep -o '"phase_ms_per_step": {[^}]*}' gpurun_out/win64_$v.log | cut -c1-140) $(grep -o '"result_checksum": [0-9]*' gpurun_out/win64_$v.log)"
done
