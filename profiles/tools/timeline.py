#!/usr/bin/env python
"""Where does an EM step go on the GPU's own clock?  Reads a rocprofv3 `--kernel-trace --output-format csv` trace of
bench.py, cuts it into passes at every k_tables launch and prints, per kernel of the pass: mean duration and mean idle
gap BEFORE it (end of the previous dispatch on the device -> its own start), plus the pass period.
Usage: timeline.py <kernel_trace.csv> [out.txt]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("void ", "")
    for cut in ("<", "("):
        if cut in n:
            n = n.split(cut)[0]
    return n


starts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_tables")]
passes = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
# steady state: drop the first passes (warm-up, profiling of every kernel) — keep the most common pass shape
shape = defaultdict(int)
for p in passes:
    shape[tuple(short(r["Kernel_Name"]) for r in p)] += 1
best = max(shape, key=shape.get)
sel = [p for p in passes if tuple(short(r["Kernel_Name"]) for r in p) == best]
out = []
out.append(f"{len(sel)} passes of shape {len(best)} dispatches (of {len(passes)} passes in the trace)")
dur = defaultdict(float)
gap = defaultdict(float)
period = 0.0
for k, p in enumerate(sel):
    for j, r in enumerate(p):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        dur[j] += (e - s) / 1e3
        if j > 0:
            gap[j] += (s - int(p[j - 1]["End_Timestamp"])) / 1e3
    period += (int(p[-1]["End_Timestamp"]) - int(p[0]["Start_Timestamp"])) / 1e3
n = len(sel)
out.append(f"{'dispatch':28s} {'dur us':>8s} {'gap before us':>14s}  vgpr  lds  grid")
tk = tg = 0.0
for j, name in enumerate(best):
    r = sel[0][j]
    out.append(f"{name:28s} {dur[j] / n:8.2f} {gap[j] / n:14.2f}  {r['VGPR_Count']:>4s} {r['LDS_Block_Size']:>5s} {r['Grid_Size_X']}x{r['Workgroup_Size_X']}")
    tk += dur[j] / n
    tg += gap[j] / n
out.append(f"sum of durations {tk:.1f} us, sum of gaps {tg:.1f} us, first start -> last end {period / n:.1f} us")
# idle time between passes
idle = 0.0
m = 0
for a, b in zip(sel[:-1], sel[1:]):
    d = (int(b[0]["Start_Timestamp"]) - int(a[-1]["End_Timestamp"])) / 1e3
    if d < 1000:
        idle += d
        m += 1
if m:
    out.append(f"idle between consecutive passes (last end -> next k_tables start): {idle / m:.1f} us")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
