#!/bin/bash
# the whole GPU suite + smoke (output uncaptured, so that a runtime abort leaves its message in the log)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu --capture=no > gpurun_out/r04_suite.log 2>&1; echo "rc=$?" >> gpurun_out/r04_suite.log
grep -v "^  File\|Extension modules\|giou_acc\|^sam2 pipeline\|^tiny\|^large\|^\.tiny\|^\.large" gpurun_out/r04_suite.log | tail -8 | cut -c1-400
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
