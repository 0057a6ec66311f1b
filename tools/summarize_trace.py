#!/usr/bin/env python
"""Summarise a rocprofv3 kernel trace (rocpd sqlite ``*_results.db`` or ``*kernel_trace.csv``).

Prints per (kernel, grid) launch count, total and average duration, sorted by total time.
Usage: python tools/summarize_trace.py <dir-or-file> [--skip-first N]
"""
import csv
import glob
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name).replace("anemoi::", "")
    name = name.replace("unsigned short", "bf16")
    return name[:100]


def rows_from_db(path):
    con = sqlite3.connect(path)
    q = "select name, duration, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size from kernels order by start"
    for name, dur, gx, wx, vg, av, lds in con.execute(q):
        yield name, dur / 1e3, gx // max(wx, 1), vg + av, lds


def rows_from_csv(path):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
            yield r["Kernel_Name"], dur, int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // wg, int(
                r.get("VGPR_Count", 0) or 0), int(r.get("LDS_Block_Size", 0) or 0)


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 45  # rows printed (0: all)
    dbs = [path] if path.endswith(".db") else glob.glob(path + "/**/*_results.db", recursive=True)
    csvs = [path] if path.endswith(".csv") else glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
    agg = defaultdict(lambda: [0, 0.0])
    for f in dbs:
        for name, dur, blocks, vgpr, lds in rows_from_db(f):
            k = (short(name), blocks, vgpr, lds)
            agg[k][0] += 1
            agg[k][1] += dur
    if not dbs:
        for f in csvs:
            for name, dur, blocks, vgpr, lds in rows_from_csv(f):
                k = (short(name), blocks, vgpr, lds)
                agg[k][0] += 1
                agg[k][1] += dur
    tot = sum(v[1] for v in agg.values())
    print(f"{'kernel':102s} {'blocks':>8s} {'vgpr':>5s} {'lds':>6s} {'n':>6s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: top or None]:
        print(f"{k[0]:102s} {k[1]:8d} {k[2]:5d} {k[3]:6d} {v[0]:6d} {v[1]:12.1f} {v[1]/v[0]:10.1f} {100*v[1]/tot:6.2f}")
    print(f"total kernel time {tot/1e3:.3f} ms over {sum(v[0] for v in agg.values())} launches")


if __name__ == "__main__":
    main()
