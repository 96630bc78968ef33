#!/bin/bash
# does Infinity-Cache residency of the weights help the decode GEMVs?  PR = number of distinct weight copies rotated through
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
for r in 12 2 1; do PB=32 PR=$r timeout 300 python tools/bench_gemv.py 2>&1 | grep -v amdgpu.ids | sed "s/^/R=$r /"; done | tee gpurun_out/mall_probe.log
