#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 (ROCm 7.2) rocpd SQLite database -> markdown/CSV summary.
Usage: python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [out.md]"""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = cur.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % of kernel time | vgpr | agpr | lds B |", "|---|---|---|---|---|---|---|---|---|---|"]
    for n, c, t, a, mn, mx, vg, ag, lds in rows:
        n = n.replace("(anonymous namespace)::", "").split("(")[0][:90]
        lines.append(f"| `{n}` | {c} | {t/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*t/total:.1f} | {vg} | {ag} | {lds} |")
    hdr = f"kernel time total {total/1e6:.2f} ms over a {((span[1]-span[0])/1e6):.2f} ms window ({len(rows)} distinct kernels)\n\n"
    txt = hdr + "\n".join(lines) + "\n"
    if out:
        open(out, "a").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])
