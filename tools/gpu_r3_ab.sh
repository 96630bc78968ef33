#!/bin/bash
# A/B of one static batch of 32 tiles: decode step + checksum per SR_DEFER_LN setting
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for v in ${VARIANTS:-0 1 0 1}; do
  SR_DEFER_LN=$v timeout 400 python bench.py --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam > gpurun_out/ab_static$v.log 2>&1
  echo "defer=$v exit $? $(grep -o '"decode_step_ms": [0-9.]*' gpurun_out/ab_static$v.log | head -1) $(grep -o '"value": [0-9.]*' gpurun_out/ab_static$v.log | head -1) $(grep -o '"result_checksum": [0-9]*' gpurun_out/ab_static$v.log)"
done
