#!/usr/bin/env python
"""Per-kernel HBM traffic from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: counters are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide coalesced read stream, so the read side is doubled (WRITE_SIZE is uncalibrated there: reported
as is).  Usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [tag]"""
import csv
import json
import sys
from collections import defaultdict


def load(path, name):
    acc, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return {k: acc[k] / cnt[k] for k in acc}, cnt


fetch, n = load(sys.argv[1], "FETCH_SIZE")
write, _ = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch, key=lambda k: -fetch[k]):
    if not k.startswith("k_"):
        continue
    rd = 2.0 * fetch[k] * 1024.0          # gfx950 correction: x2
    wr = write.get(k, 0.0) * 1024.0
    out[k] = {"launches": n[k], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
if len(sys.argv) > 4:      # the snapshot's tag: bench.py finds the SQ / fp64 passes of the SAME snapshot through it (profiles/<tag>_pmc_*.json)
    out["_meta"] = {"tag": sys.argv[4]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
out.pop("_meta", None)
for k, v in out.items():
    print(f"{k:28s} read {v['read_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB")
