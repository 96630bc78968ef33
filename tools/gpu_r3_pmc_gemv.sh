#!/bin/bash
cd "$(dirname "$0")/.."; R=$PWD; mkdir -p gpurun_out/pmc_r3; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python $R/tools/probe_r2.py gemv > $R/gpurun_out/pmc_r3/gemv_$c.log 2>&1
  python $R/tools/rocpd_pmc.py "$(find /tmp/pmc_$c -name '*.db' | head -1)" $R/gpurun_out/pmc_r3/gemv_$c.json > /dev/null 2>> $R/gpurun_out/pmc_r3/gemv_$c.log
done
cd $R; python tools/gemv_traffic.py gpurun_out/pmc_r3/gemv_FETCH_SIZE.json gpurun_out/pmc_r3/gemv_WRITE_SIZE.json gpurun_out/r03_pmc_gemv_traffic.json | tail -30
