#!/usr/bin/env python
"""Average per-launch value of every counter in a rocprofv3 --pmc counter_collection.csv, per kernel.
Usage: pmc_counters.py <counter_collection.csv> [out.json]"""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not k.startswith("k_"):
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
out = {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
names = sorted({c for k in out for c in out[k]})
print("kernel".ljust(22) + "".join(n[-18:].rjust(19) for n in names))
for k in sorted(out):
    print(k[:22].ljust(22) + "".join(f"{out[k].get(n, 0):19.4g}" for n in names))
