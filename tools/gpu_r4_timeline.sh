#!/bin/bash
# tile timeline of the 256-tile GEMM: needs socioreasoner_amd/libsocior_timing.so (gemm256.hip built with -DSR_G256_TIMING, linked with the
# product objects: see the hipcc lines in DESIGN.md section 4 "where a tile's time goes")
mkdir -p gpurun_out
python tools/probe_gemm256_timeline.py > gpurun_out/r04_gemm256_timeline.jsonl 2> gpurun_out/r04_gemm256_timeline.err || tail -5 gpurun_out/r04_gemm256_timeline.err
cat gpurun_out/r04_gemm256_timeline.jsonl
