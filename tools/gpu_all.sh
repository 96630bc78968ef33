#!/bin/bash
# parity stages, then the profiled bench (rocprofv3 kernel trace) in one gpurun call
cd "$(dirname "$0")/.."
BENCH=0 bash tools/gpu_check.sh
bash tools/gpu_profile.sh
