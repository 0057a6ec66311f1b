#!/usr/bin/env python
"""Idle time between the kernels of a rocprofv3 kernel trace (``*kernel_trace.csv``): how much of a step is launch gaps.

    python tools/trace_gaps.py <dir-or-csv> [--last-frac 0.5]

Takes the kernels of the last ``--last-frac`` of the trace (steady state), sorts them by start time and prints the busy
time (union of the kernel intervals), the idle time between them, the gap histogram and the kernels behind the longest gaps."""
import csv
import glob
import sys
from collections import Counter, defaultdict


def main():
    path = sys.argv[1]
    frac = float(sys.argv[sys.argv.index("--last-frac") + 1]) if "--last-frac" in sys.argv else 0.5
    files = [path] if path.endswith(".csv") else glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[int(len(rows) * (1 - frac)):]
    if not rows:
        print("no kernels")
        return
    span = rows[-1][1] - rows[0][0]
    busy, cur_end = 0, rows[0][0]
    gaps, after = [], defaultdict(lambda: [0, 0])
    for (s, e, name), prev in zip(rows, [None] + rows[:-1]):
        if s > cur_end:
            if prev is not None:
                g = s - cur_end
                gaps.append(g)
                key = prev[2].split("(")[0][-60:]
                after[key][0] += 1
                after[key][1] += g
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    idle = span - busy
    print(f"{len(rows)} kernels over {span / 1e6:.3f} ms: busy {busy / 1e6:.3f} ms, idle {idle / 1e6:.3f} ms ({100 * idle / span:.1f} %)")
    hist = Counter()
    for g in gaps:
        b = "<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-10us" if g < 10000 else "10-50us" if g < 50000 else ">50us"
        hist[b] += 1
    print("gaps:", {k: hist[k] for k in ("<1us", "1-2us", "2-4us", "4-10us", "10-50us", ">50us")},
          f"median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us" if gaps else "")
    print("idle time by the kernel in FRONT of the gap:")
    for k, (n, g) in sorted(after.items(), key=lambda kv: -kv[1][1])[:15]:
        print(f"  {k:62s} n={n:5d} total {g / 1e3:9.1f} us  avg {g / n / 1e3:6.2f} us")


if __name__ == "__main__":
    main()
