#!/bin/bash
# Round 4: the measurements committed under profiles/ -- kernel traces (static batch 32, batch 1, the HEADLINE continuous run if rocprofv3 survives its
# CU-masked streams, 64 / 128 rows static), GEMV HBM traffic by PMC (separate FETCH_SIZE / WRITE_SIZE passes), then the default bench line last.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/prof4 gpurun_out/pmc_r4
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof4 -o s32 -- python $R/bench.py --static --steps 2 --warmup 1 --no-cpu-baseline --no-latency > $R/gpurun_out/r04_prof_s32.log 2>&1; echo "trace s32 exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof4 -o b1 -- python $R/bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-sam > $R/gpurun_out/r04_prof_b1.log 2>&1; echo "trace b1 exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof4 -o s128 -- python $R/bench.py --batch 128 --static --steps 2 --warmup 1 --no-cpu-baseline --no-latency > $R/gpurun_out/r04_prof_s128.log 2>&1; echo "trace s128 exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof4 -o c32 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency > $R/gpurun_out/r04_prof_c32.log 2>&1; echo "trace c32 (headline, CU-masked streams) exit $?"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p -- python $R/tools/probe_r2.py gemv > $R/gpurun_out/pmc_r4/gemv_$c.log 2>&1
  python $R/tools/rocpd_pmc.py "$(find /tmp/pmc_$c -name '*.db' | head -1)" $R/gpurun_out/pmc_r4/gemv_$c.json > /dev/null 2>> $R/gpurun_out/pmc_r4/gemv_$c.log
done
cd $R
python tools/gemv_traffic.py gpurun_out/pmc_r4/gemv_FETCH_SIZE.json gpurun_out/pmc_r4/gemv_WRITE_SIZE.json gpurun_out/r04_pmc_gemv_traffic.json | tail -6
for n in s32 b1 s128 c32; do
  DB=$(find gpurun_out/prof4 -name "${n}_results.db" | head -1)
  rm -f gpurun_out/r04_bench_${n}_kernel_stats.md
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB gpurun_out/r04_bench_${n}_kernel_stats.md > /dev/null && head -12 gpurun_out/r04_bench_${n}_kernel_stats.md | cut -c1-150
done
python tools/gemv_in_situ.py gpurun_out/r04_bench_s32_kernel_stats.md gpurun_out/r04_bench_b1_kernel_stats.md gpurun_out/r04_gemv_in_situ.json
cp gpurun_out/r04_gemv_in_situ.json profiles/r04_gemv_in_situ.json
cp gpurun_out/r04_pmc_gemv_traffic.json profiles/r04_pmc_gemv_traffic.json
SR_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 1 --warmup 0 --waves 1 --no-latency --no-cpu-baseline --no-sam 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 gloo ranks, continuous:', d['value'], d['n_gpus'], d['config']['exchange']['nranks'])"
OUT=/tmp/example_out SOCIOSEG_NUM_SAMPLES=32 timeout 600 python tools/run_example_small.py 2>/dev/null | tail -1 | cut -c1-600
timeout 1200 python bench.py > gpurun_out/r04_bench_default.log 2> gpurun_out/r04_bench_default.err; echo "bench exit $?"
tail -n 1 gpurun_out/r04_bench_default.log > gpurun_out/r04_bench_default_line.json
cut -c1-300 gpurun_out/r04_bench_default_line.json
rm -rf gpurun_out/prof4
