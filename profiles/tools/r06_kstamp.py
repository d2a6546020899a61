#!/usr/bin/env python3
"""Start-to-start distances of the default pass's four launches (a -DHF_KSTAMP build; HF_KSTAMP_FILE=<file>): the wall clock (100 MHz) of each
kernel's block 0, a ring of 64 passes.  Usage: r06_kstamp.py <file> [first_pass last_pass]"""
import struct, sys
import numpy as np
raw = open(sys.argv[1], "rb").read()
n = len(raw) // (257 * 8)
best = None
for i in range(n):
    h = struct.unpack("<257Q", raw[i * 257 * 8:(i + 1) * 257 * 8])
    if best is None or h[256] > best[256]:
        best = h
last = best[256]
print("passes counted by the device: %d (the ring holds the last 64)" % last)
rows = {}
for p in range(max(1, last - 62), last + 1):
    rows[p] = [best[(p & 63) * 4 + k] for k in range(4)]
names = ["k_tables -> k_seg_fb", "k_seg_fb -> k_pair_sums", "k_pair_sums -> k_row_stats", "k_row_stats -> next k_tables", "k_tables -> next k_tables"]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else min(rows)
hi = int(sys.argv[3]) if len(sys.argv) > 3 else max(rows) - 1
if lo < 0: lo += last      # (negative: counted back from the last pass)
if hi < 0: hi += last
out = []
for p in range(lo, hi + 1):
    if p not in rows or p + 1 not in rows: continue
    t = rows[p]; nt = rows[p + 1][0]
    d = [(t[1] - t[0]) / 100.0, (t[2] - t[1]) / 100.0, (t[3] - t[2]) / 100.0, (nt - t[3]) / 100.0, (nt - t[0]) / 100.0]
    out.append((p, d))
for p, d in out:
    print("pass %3d  " % p + "  ".join("%6.2f" % v for v in d))
a = np.array([d for _, d in out])
print("start-to-start, us (median | mean) over passes %d..%d:" % (lo, hi))
for k, nm in enumerate(names):
    print("  %-30s %6.2f | %6.2f" % (nm, np.median(a[:, k]), a[:, k].mean()))
