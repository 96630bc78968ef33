#!/bin/bash
# the whole -m gpu suite + smoke, as the driver runs them at round end
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/suite.log 2>&1; echo "suite exit $?" | tee gpurun_out/suite_summary.txt
tail -6 gpurun_out/suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/suite_summary.txt
tail -3 gpurun_out/smoke.log
