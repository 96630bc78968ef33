#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for cfg in "4 16" "0 16" "4 8" "4 4" "8 4" "0 4"; do
  set -- $cfg
  SR_MIN_GROUP=$1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-sam --poll-ragged $2 > gpurun_out/poll.log 2>&1
  echo "min_group=$1 poll-ragged=$2 exit $? $(grep -o '"value": [0-9.]*' gpurun_out/poll.log | head -1) $(grep -o '"continuous_tiles_per_s": [0-9.]*' gpurun_out/poll.log) $(grep -o '"continuous_decode_steps": [0-9]*' gpurun_out/poll.log) $(grep -o '"gain": [0-9.]*' gpurun_out/poll.log)"
done
