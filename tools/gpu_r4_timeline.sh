#!/bin/bash
# tile / launch timelines: need socioreasoner_amd/libsocior_timing.so (gemm256.hip with -DSR_G256_TIMING and gemv.hip with -DSR_GEMV_TIMING,
# linked with the product objects: the hipcc lines are in tools/experiments/README.md)
mkdir -p gpurun_out
python tools/probe_gemm256_timeline.py > gpurun_out/r04_gemm256_timeline.jsonl 2> gpurun_out/r04_gemm256_timeline.err || tail -5 gpurun_out/r04_gemm256_timeline.err
python tools/probe_gemv_timeline.py > gpurun_out/r04_gemv_timeline.jsonl 2> gpurun_out/r04_gemv_timeline.err || tail -5 gpurun_out/r04_gemv_timeline.err
cat gpurun_out/r04_gemv_timeline.jsonl
