#!/usr/bin/env python
"""Summarise rocprofv3 --pmc output (rocpd sqlite or csv): per kernel, mean counter value per dispatch."""
import glob
import sqlite3
import sys
from collections import defaultdict

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for db in glob.glob(path + "/**/*_results.db", recursive=True):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    if "counters_collection" not in tabs:
        continue
    cols = [d[1] for d in con.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    agg = defaultdict(lambda: [0.0, 0])
    q = f"select {name_col}, counter_name, value from counters_collection"
    for kname, cname, val in con.execute(q):
        if pat and pat not in kname:
            continue
        a = agg[(kname[:70], cname)]
        a[0] += float(val)
        a[1] += 1
    for (k, c), (tot, n) in sorted(agg.items()):
        print(f"{k:70s} {c:28s} n={n:4d} mean={tot / n:16.1f}")
