#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/samprof -o sam -- python $R/tools/prof_sam2_encoder.py > $R/gpurun_out/sam_prof.log 2>&1
cd $R
DB=$(find /tmp/samprof -name 'sam_results.db' | head -1)
rm -f gpurun_out/r03_sam2_encoder_kernel_stats.md
python tools/rocpd_stats.py $DB gpurun_out/r03_sam2_encoder_kernel_stats.md > /dev/null
python tools/rocpd_by_grid.py $DB gpurun_out/r03_sam2_encoder_by_grid.md 45 | cut -c1-170
head -14 gpurun_out/r03_sam2_encoder_kernel_stats.md | cut -c1-150
