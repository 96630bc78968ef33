#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/samdprof -o samd -- python $R/tools/prof_sam2_decoder.py > $R/gpurun_out/sam_dec_prof.log 2>&1; echo "trace exit $?"
cd $R
DB=$(find /tmp/samdprof -name 'samd_results.db' | head -1)
rm -f gpurun_out/r03_sam2_decoder_kernel_stats.md
python tools/rocpd_stats.py $DB gpurun_out/r03_sam2_decoder_kernel_stats.md > /dev/null
head -24 gpurun_out/r03_sam2_decoder_kernel_stats.md | cut -c1-150
