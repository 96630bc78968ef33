#!/usr/bin/env python3
"""Per-kernel PMC averages from a rocprofv3 (ROCm 7.2) rocpd database: for every kernel and counter, the counter summed over
its instances (XCDs / SEs) per dispatch, averaged over the dispatches, next to the average dispatch duration.
Usage: python tools/rocpd_pmc.py <results.db> [out.json]"""
import json
import sqlite3
import sys


def main(path, out=None):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, dispatch_id, counter_name, sum(counter_value), max(duration) from pmc_events "
                       "group by name, dispatch_id, counter_name").fetchall()
    acc = {}
    for name, did, cn, val, dur in rows:
        k = acc.setdefault(name, {}).setdefault(cn, [0.0, 0, 0.0])
        k[0] += val
        k[1] += 1
        k[2] += dur
    res = {}
    for name, cs in acc.items():
        short = name.replace("(anonymous namespace)::", "").split("(")[0][:100]
        res[short] = {cn: {"per_dispatch": v[0] / v[1], "dispatches": v[1], "avg_us": v[2] / v[1] / 1e3} for cn, v in cs.items()}
    txt = json.dumps(res, indent=1)
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])
