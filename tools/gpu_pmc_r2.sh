#!/bin/bash
# Round-2 PMC passes (separate runs per counter group, kernel trace only -- the guide's rule).  Output: gpurun_out/pmc_r2/*.json
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/pmc_r2
export TMPDIR=/tmp
cd /tmp
run() {  # name, probe arg, counters...
  name=$1; arg=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$name -o p -- python $R/tools/probe_r2.py $arg > $R/gpurun_out/pmc_r2/$name.log 2>&1
  db=$(find /tmp/pmc_$name -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py "$db" $R/gpurun_out/pmc_r2/$name.json > /dev/null 2>> $R/gpurun_out/pmc_r2/$name.log
  rm -rf /tmp/pmc_$name
}
run gemv_fetch gemv FETCH_SIZE
run gemv_write gemv WRITE_SIZE
run gemm_fetch gemm FETCH_SIZE
run gemm_mfma gemm SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
run gemm_waves gemm SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run gemm_lds gemm SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16
ls -la $R/gpurun_out/pmc_r2
