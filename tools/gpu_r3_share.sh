#!/bin/bash
# admission CU share (of 8 per shader engine): headline and ragged phase
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for c in ${SHARES:-auto 2 4 5}; do
  timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sam --admit-cus $c > gpurun_out/share.log 2>&1
  echo "admit-cus=$c exit $? $(grep -o '"value": [0-9.]*' gpurun_out/share.log | head -1) ragged $(grep -o '"continuous_tiles_per_s": [0-9.]*' gpurun_out/share.log) $(grep -o '"continuous_decode_steps": [0-9]*' gpurun_out/share.log)"
done
