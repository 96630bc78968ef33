#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "gemm or full_size or batch_variants" > gpurun_out/r04_o_tests.log 2>&1; tail -3 gpurun_out/r04_o_tests.log
bash tools/gpu_r4_timeline.sh | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'span', d['span_us'], 'pro', d['prologue_us'], 'loop', d['k_loop_us'], 'epi', d['epilogue_issue_us'], 'block', d['block_us'])
"
timeout 900 python bench.py --static --steps 2 --warmup 1 --no-latency --no-cpu-baseline --no-sam --no-more-rows > gpurun_out/r04_o_static.log 2> gpurun_out/r04_o_static.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04_o_static.log").read().strip().splitlines()[-1])
print(d["value"], {k: d["phase_ms_per_step"].get(k) for k in ("vit", "prefill", "decode", "vit_mfma_frac", "prefill_mfma_frac", "forward_mfma_frac")})
PY
