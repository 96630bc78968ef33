#!/usr/bin/env python3
"""Idle time BETWEEN consecutive kernels of the decode graph, from a rocprofv3 rocpd database (kernel trace of `bench.py --static`): dispatches in
start order, gap = next start - previous end, kept when both kernels belong to the decode step (GEMV / decode attention / RMSNorm-row / step
kernels) and the gap is below 50 us.  Says how much of a decode step is not inside any kernel.
Usage: python tools/rocpd_gaps.py <results.db> [out.json]"""
import json
import sqlite3
import sys

import numpy as np

DECODE = ("k_gemv", "k_attn_dec", "k_rmsnorm_row", "k_step")


def main(path, out=None):
    db = sqlite3.connect(path)
    rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    names = [r[0] for r in rows]
    st = np.array([r[1] for r in rows], dtype=np.int64)
    en = np.array([r[2] for r in rows], dtype=np.int64)
    dec = np.array([any(k in n for k in DECODE) for n in names])
    gaps, by_pair, dur = [], {}, []
    for i in range(len(rows) - 1):
        if dec[i] and dec[i + 1]:
            g = (st[i + 1] - en[i]) / 1e3
            if g < 50:
                gaps.append(g)
                key = names[i].split("(")[0].replace("(anonymous namespace)::", "")[-40:] + " -> " + names[i + 1].split("(")[0].replace("(anonymous namespace)::", "")[-40:]
                by_pair.setdefault(key, []).append(g)
                dur.append((en[i] - st[i]) / 1e3)
    gaps = np.array(gaps)
    res = {"decode_kernel_pairs": int(len(gaps)), "gap_us_median": round(float(np.median(gaps)), 3), "gap_us_mean": round(float(gaps.mean()), 3),
           "gap_us_p10_p90": [round(float(np.percentile(gaps, 10)), 3), round(float(np.percentile(gaps, 90)), 3)],
           "kernel_us_mean": round(float(np.mean(dur)), 3), "gap_share_of_decode_time": round(float(gaps.sum() / (gaps.sum() + np.sum(dur))), 4),
           "by_pair_median_us": {k: round(float(np.median(v)), 3) for k, v in sorted(by_pair.items(), key=lambda kv: -len(kv[1]))[:12]}}
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
