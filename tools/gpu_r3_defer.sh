#!/bin/bash
# deferred-row-scale RMSNorm in the batch-32 decode layer: A/B of one static batch of 32 tiles + parity tests with the switch on
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/summary_defer.txt
for v in ${VARIANTS:-0 1 0 1}; do
  SR_DEFER_LN=$v timeout 400 python bench.py --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam > gpurun_out/defer_static$v.log 2>&1
  echo "defer=$v exit $? $(grep -o '"decode_step_ms": [0-9.]*' gpurun_out/defer_static$v.log | head -1) $(grep -o '"value": [0-9.]*' gpurun_out/defer_static$v.log | head -1) $(grep -o '"result_checksum": [0-9]*' gpurun_out/defer_static$v.log)" | tee -a gpurun_out/summary_defer.txt
done
SR_DEFER_LN=${TESTV:-1} timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x -k "${TESTK:-truth or decode or batch or engine or serving or pipeline}" > gpurun_out/defer_tests.log 2>&1
echo "tests exit $?" | tee -a gpurun_out/summary_defer.txt
tail -15 gpurun_out/defer_tests.log
