#!/bin/bash
# round-4 final check: the whole GPU suite twice (output uncaptured), smoke, then the default bench line
mkdir -p gpurun_out
for rep in 1 2; do
  timeout 2400 python -m pytest tests -x -q -m gpu --capture=no > gpurun_out/r04_final_suite$rep.log 2>&1; echo "suite $rep rc=$?"
  grep -E "passed|failed|error|Aborted|Fatal" gpurun_out/r04_final_suite$rep.log | tail -3
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py > gpurun_out/r04_bench_default.log 2> gpurun_out/r04_bench_default.err; echo "bench exit $?"
tail -n 1 gpurun_out/r04_bench_default.log > gpurun_out/r04_bench_default_line.json
cut -c1-200 gpurun_out/r04_bench_default_line.json
