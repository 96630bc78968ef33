#!/usr/bin/env python3
"""profiles/r03_gemv_in_situ.json from the two kernel-stats tables (static batch 32, batch 1): the launch-count-weighted average duration of
every k_gemv / k_gemv32 launch inside the decode steps -- the in-situ figure bench.py prints beside its replayed roofline number."""
import json
import re
import sys


def table(path):
    rows = {}
    for l in open(path):
        m = re.match(r"\| `([^`]+)` \| (\d+) \| ([0-9.]+) \| ([0-9.]+) \|", l)
        if m and re.search(r"k_gemv(32)?I", m.group(1)):
            rows[m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    calls = sum(c for c, _, _ in rows.values())
    total_us = sum(t for _, t, _ in rows.values()) * 1e3
    return round(total_us / calls, 2), {re.sub(r"^_ZN12_GLOBAL__N_1", "", k)[:40]: a for k, (c, t, a) in rows.items()}


s32, b1, out = sys.argv[1:4]
a32, k32 = table(s32)
a1, k1 = table(b1)
json.dump({"source": f"rocprofv3 --kernel-trace --stats of bench.py --static (batch 32) and --batch 1: average duration of ALL k_gemv / k_gemv32 launches inside the "
                     f"decode steps ({s32}, {b1}; tools/gemv_in_situ.py)", "batch32": a32, "batch1": a1, "per_kernel_avg_us_batch32": k32, "per_kernel_avg_us_batch1": k1},
          open(out, "w"), indent=1)
print(a32, a1)
