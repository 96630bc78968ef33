#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench command; summaries are copied to profiles/ by hand afterwards.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
tail -n 2 gpurun_out/prof_bench.log
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -30 "$f"
# keep the merge small: drop the raw trace if it is big
find gpurun_out/prof -name "*kernel_trace.csv" -size +30M -delete
