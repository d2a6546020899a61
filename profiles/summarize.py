#!/usr/bin/env python
"""Turn a rocprofv3 rocpd sqlite result (`*_results.db`, --kernel-trace --stats) into the per-kernel
summary CSV that is committed under profiles/.  Usage: summarize.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for r in rows:
        w.writerow([r[0], r[1], f"{r[2]:.3f}", f"{r[3]:.3f}", f"{r[4]:.3f}"])
print(f"{len(rows)} kernels -> {sys.argv[2]}")
