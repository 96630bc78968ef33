#!/bin/bash
# rocprofv3 of the bench command: (1) kernel trace, (2)+(3) separate PMC passes for HBM read / write bytes.
# Output: rocpd SQLite databases under gpurun_out/prof/; tools/rocpd_stats.py turns them into the summaries in profiles/.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/prof_bench.log 2>&1
echo "rocprof trace exit $?"
if [ "${PMC:-0}" = "1" ]; then
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof -o pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/prof_pmc_fetch.log 2>&1
echo "rocprof pmc fetch exit $?"
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof -o pmc_write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/prof_pmc_write.log 2>&1
echo "rocprof pmc write exit $?"
fi
cd $R
grep -o '"value": [0-9.]*, "unit": "tiles/s"' gpurun_out/prof_bench.log
ls -la gpurun_out/prof
