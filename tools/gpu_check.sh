#!/bin/bash
# One gpurun call: staged GPU checks, each stage in its own process with its own timeout; logs under gpurun_out/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
stage() { name=$1; shift; echo "=== $name"; timeout "$T" "$@" > gpurun_out/$name.log 2>&1; echo "$name exit $?" | tee -a gpurun_out/summary.txt; tail -n ${TAILN:-15} gpurun_out/$name.log; }
: > gpurun_out/summary.txt
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit" | head -4 >> gpurun_out/summary.txt
nproc >> gpurun_out/summary.txt
T=900 stage ops_tiny python -m pytest tests/test_gpu_parity.py -q -m gpu -k "not (truedim or full_size or ragged)" -p no:cacheprovider
T=900 stage truedim python -m pytest tests/test_gpu_parity.py -q -m gpu -k "truedim or full_size or ragged" -p no:cacheprovider
T=600 stage pipeline python -m pytest tests/test_gpu_pipeline.py -q -m gpu -p no:cacheprovider
T=300 stage smoke python -c "import __graft_entry__ as g; g.smoke()"
if [ "${BENCH:-1}" = "1" ]; then
T=900 TAILN=3 stage bench python bench.py --steps 2 --warmup 1
fi
cat gpurun_out/summary.txt
