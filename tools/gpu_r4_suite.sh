#!/bin/bash
# the whole GPU suite + smoke
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r04_suite.log
tail -5 gpurun_out/r04_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
