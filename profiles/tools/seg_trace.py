#!/usr/bin/env python
"""k_seg_fb from a -DHF_SEG_TRACE build (profiles/tools/seg_trace.sh; HF_SEG_TRACE_FILE=<file>): where a forward step goes
(s_memtime, wavefront 0 of every workgroup), how long workgroups live and how they spread over the CUs (s_memrealtime, HW_ID)."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 10).astype(np.int64)
L = t[:, 4].astype(float)
print("workgroups", len(t), " windows per lane: mean %.2f" % L.mean(), " windows per segment: mean %.0f" % t[:, 5].mean())
for k, nm in enumerate(["wait for the DMA + read the rows out of LDS", "deferred stores", "issue the next step's row fetch", "arithmetic"]):
    print("forward step, %-46s %6.0f cycles" % (nm, (t[:, k] / L).mean()))
s, e = t[:, 6], t[:, 7]
t0 = s.min()
life = (e - s) / 100.0
print("kernel span %.1f us; workgroup life: p10 %.1f, median %.1f, p90 %.1f us; sum of lives / span = %.0f workgroups busy on average"
      % ((e.max() - t0) / 100.0, np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.sum() / ((e.max() - t0) / 100.0)))
print("start (us after the first):", [round((np.percentile(s, q) - t0) / 100.0, 1) for q in (50, 90, 100)],
      " end:", [round((np.percentile(e, q) - t0) / 100.0, 1) for q in (0, 10, 50, 90, 100)])
hw, xcc = t[:, 8], t[:, 9]
key = (xcc & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 20 + ((hw >> 8) & 0xf)
u, c = np.unique(key, return_counts=True)
print("CUs used", len(u), " workgroups per CU: histogram", dict(enumerate(np.bincount(c))))
