#!/bin/bash
# Round 3, call A: attention v2 -- bit identity, A/B timing of one static batch of 32 tiles, kernel trace, float32-truth tests.
cd "$(dirname "$0")/.."
R=$PWD
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
: > gpurun_out/summary_a.txt
stage() { name=$1; shift; echo "=== $name"; timeout "$T" "$@" > gpurun_out/$name.log 2>&1; echo "$name exit $?" | tee -a gpurun_out/summary_a.txt; tail -n ${TAILN:-12} gpurun_out/$name.log; }
T=600 stage a_bitid python -m pytest tests/test_gpu_round3.py -q -m gpu -k attention_v2 -p no:cacheprovider -x
for v in 0 1 0 1; do
  SR_ATTN2=$v T=400 TAILN=1 stage a_static_attn$v python bench.py --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline
  grep -o '"phase_ms_per_step": {[^}]*}' gpurun_out/a_static_attn$v.log | tee -a gpurun_out/summary_a.txt
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r3a_static -- python $R/bench.py --static --steps 1 --warmup 1 --no-latency --no-cpu-baseline > $R/gpurun_out/a_prof.log 2>&1
echo "rocprof exit $?" | tee -a $R/gpurun_out/summary_a.txt
cd $R
DB=$(find gpurun_out/prof -name "r3a_static*results.db" | head -1); python tools/rocpd_stats.py $DB gpurun_out/r3a_static_kernel_stats.md | head -30
T=900 TAILN=25 stage a_truth python -m pytest tests/test_gpu_round3.py -q -m gpu -k truth -p no:cacheprovider
cat gpurun_out/summary_a.txt
