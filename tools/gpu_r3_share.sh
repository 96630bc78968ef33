#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for c in auto 2 4 auto; do
  timeout 500 python bench.py --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam --admit-cus $c > gpurun_out/share.log 2>&1
  echo "admit-cus=$c exit $? $(grep -o '"value": [0-9.]*' gpurun_out/share.log | head -1) $(grep -o '"decode_step_ms_shared": [0-9.]*' gpurun_out/share.log | head -1) $(grep -o '"steps_shared": [0-9]*' gpurun_out/share.log | head -1)"
done
