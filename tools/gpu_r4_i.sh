#!/bin/bash
# round 4: kernel trace of the float32 SAM2 encoder (8 tiles per pass) by kernel and by grid; the rows tests after the abort fix
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/prof4
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -k "above_32 or abort" 2>&1 | tail -4
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof4 -o samf32 -- python $R/tools/prof_sam2_encoder.py f32 > $R/gpurun_out/r04_prof_samf32.log 2>&1; echo "trace exit $?"
cd $R
DB=$(find gpurun_out/prof4 -name "samf32_results.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r04_sam2_f32_encoder_kernel_stats.md > /dev/null
python tools/rocpd_by_grid.py $DB gpurun_out/r04_sam2_f32_encoder_by_grid.md 30 > /dev/null
head -16 gpurun_out/r04_sam2_f32_encoder_kernel_stats.md | cut -c1-160
head -34 gpurun_out/r04_sam2_f32_encoder_by_grid.md | cut -c1-150
rm -rf gpurun_out/prof4
