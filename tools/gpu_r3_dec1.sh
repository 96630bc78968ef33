#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "one_launch or decode_attention" 2>&1 | tail -5
for v in 0 1 0 1; do
  SR_ATTN_DEC1=$v timeout 400 python bench.py --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam > gpurun_out/dec1_$v.log 2>&1
  echo "dec1=$v exit $? $(grep -o '"decode_step_ms": [0-9.]*' gpurun_out/dec1_$v.log | head -1) $(grep -o '"value": [0-9.]*' gpurun_out/dec1_$v.log | head -1) $(grep -o '"result_checksum": [0-9]*' gpurun_out/dec1_$v.log)"
done
