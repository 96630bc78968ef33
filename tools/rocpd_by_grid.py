#!/usr/bin/env python3
"""Per-(kernel, grid) statistics from a rocprofv3 rocpd database: which SHAPES of a kernel family cost the time.
Usage: python tools/rocpd_by_grid.py results.db [out.md] [top_n]"""
import sqlite3
import sys


def main(path, out=None, top=40):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    gx, gy, gz = [c for c in ("grid_size_x", "grid_x") if c in cols][0], [c for c in ("grid_size_y", "grid_y") if c in cols][0], [c for c in ("grid_size_z", "grid_z") if c in cols][0]
    wx = [c for c in ("workgroup_size_x", "workgroup_x") if c in cols][0]
    rows = cur.execute(
        f"select s.kernel_name, d.{gx}, d.{gy}, d.{gz}, d.{wx}, count(*), sum(d.end-d.start), avg(d.end-d.start) "
        f"from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1, 2, 3, 4 order by 7 desc").fetchall()
    total = sum(r[6] for r in rows)
    lines = ["| kernel | grid (work-items) | block | calls | total ms | avg us | % |", "|---|---|---|---|---|---|---|"]
    for n, x, y, z, w, c, t, a in rows[:int(top)]:
        n = n.replace("(anonymous namespace)::", "").split("(")[0][:70]
        lines.append(f"| `{n}` | {x} x {y} x {z} | {w} | {c} | {t/1e6:.3f} | {a/1e3:.2f} | {100*t/total:.1f} |")
    txt = f"kernel time total {total/1e6:.2f} ms\n\n" + "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])
