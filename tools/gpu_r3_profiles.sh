#!/bin/bash
# Round 3: the measurements committed under profiles/ -- default bench line, kernel traces (continuous + static batch 32, batch 1), attention PMC.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/prof3
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/r03_bench_default.log 2>&1; echo "bench exit $?"
tail -n 1 gpurun_out/r03_bench_default.log > gpurun_out/r03_bench_default_line.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof3 -o c32 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency > $R/gpurun_out/r03_prof_c32.log 2>&1; echo "trace c32 exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof3 -o s32 -- python $R/bench.py --static --steps 2 --warmup 1 --no-cpu-baseline --no-latency > $R/gpurun_out/r03_prof_s32.log 2>&1; echo "trace s32 exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof3 -o b1 -- python $R/bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r03_prof_b1.log 2>&1; echo "trace b1 exit $?"
cd $R
for n in c32 s32 b1; do
  DB=$(find gpurun_out/prof3 -name "${n}_results.db" | head -1)
  rm -f gpurun_out/r03_bench_${n}_kernel_stats.md
  python tools/rocpd_stats.py $DB gpurun_out/r03_bench_${n}_kernel_stats.md > /dev/null
  head -12 gpurun_out/r03_bench_${n}_kernel_stats.md
done
bash tools/gpu_pmc_attn.sh > gpurun_out/r03_pmc_attn.log 2>&1; tail -n 24 gpurun_out/r03_pmc_attn.log | grep prefill2
rm -rf gpurun_out/prof3
