#!/bin/bash
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
SOCIOSEG_NUM_SAMPLES=${N:-64} NEW_TOKENS=${NEW:-128} timeout 900 python tools/run_example_small.py > gpurun_out/example_small.log 2>&1; echo "example exit $?"
tail -5 gpurun_out/example_small.log | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_sam2.py tests/test_gpu_pipeline.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4
