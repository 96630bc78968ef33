#!/bin/bash
# PMC passes over the prefill attention kernels (separate runs per counter group, kernel trace only).  Output: gpurun_out/pmc_attn/*.json
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/pmc_attn
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$name -o p -- python $R/tools/probe_attn.py > $R/gpurun_out/pmc_attn/$name.log 2>&1
  db=$(find /tmp/pmc_$name -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py "$db" $R/gpurun_out/pmc_attn/$name.json > /dev/null 2>> $R/gpurun_out/pmc_attn/$name.log
  rm -rf /tmp/pmc_$name
}
run waves SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES
run valu SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run lds SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run vmem SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run wait SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run fetch FETCH_SIZE
ls -la $R/gpurun_out/pmc_attn
python - <<PY
import json,glob
for f in sorted(glob.glob("$R/gpurun_out/pmc_attn/*.json")):
    d=json.load(open(f))
    for k,v in d.items():
        if "attn_prefill" in k:
            print(f.split("/")[-1], k[:60], {c:(round(x["per_dispatch"]), round(x["avg_us"],1)) for c,x in v.items()})
PY
