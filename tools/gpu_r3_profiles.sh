#!/bin/bash
# Round 3: the measurements committed under profiles/ -- default bench line, kernel traces (static batch 32, batch 1, SAM2 encoder), attention PMC.
# (a kernel trace of the CONTINUOUS run segfaults rocprofv3 on this image -- two CU-masked streams; the static batch runs the same kernels)
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/prof3
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof3 -o s32 -- python $R/bench.py --static --steps 2 --warmup 1 --no-cpu-baseline --no-latency > $R/gpurun_out/r03_prof_s32.log 2>&1; echo "trace s32 exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof3 -o b1 -- python $R/bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-sam > $R/gpurun_out/r03_prof_b1.log 2>&1; echo "trace b1 exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof3 -o sam -- python $R/tools/prof_sam2_encoder.py > $R/gpurun_out/r03_prof_sam.log 2>&1; echo "trace sam exit $?"
cd $R
for n in s32 b1; do
  DB=$(find gpurun_out/prof3 -name "${n}_results.db" | head -1)
  rm -f gpurun_out/r03_bench_${n}_kernel_stats.md
  python tools/rocpd_stats.py $DB gpurun_out/r03_bench_${n}_kernel_stats.md > /dev/null
  head -12 gpurun_out/r03_bench_${n}_kernel_stats.md | cut -c1-140
done
DB=$(find gpurun_out/prof3 -name "sam_results.db" | head -1)
rm -f gpurun_out/r03_sam2_encoder_kernel_stats.md
python tools/rocpd_stats.py $DB gpurun_out/r03_sam2_encoder_kernel_stats.md > /dev/null
python tools/rocpd_by_grid.py $DB gpurun_out/r03_sam2_encoder_by_grid.md 40 > /dev/null
python tools/gemv_in_situ.py gpurun_out/r03_bench_s32_kernel_stats.md gpurun_out/r03_bench_b1_kernel_stats.md gpurun_out/r03_gemv_in_situ.json
# the default line LAST: it reads profiles/r03_gemv_in_situ.json, refreshed from the traces above
cp gpurun_out/r03_gemv_in_situ.json profiles/r03_gemv_in_situ.json
timeout 900 python bench.py > gpurun_out/r03_bench_default.log 2>&1; echo "bench exit $?"
tail -n 1 gpurun_out/r03_bench_default.log > gpurun_out/r03_bench_default_line.json
cut -c1-400 gpurun_out/r03_bench_default_line.json
if [ -z "$SKIP_PMC" ]; then bash tools/gpu_pmc_attn.sh > gpurun_out/r03_pmc_attn.log 2>&1; tail -n 24 gpurun_out/r03_pmc_attn.log | grep prefill2; fi
rm -rf gpurun_out/prof3
