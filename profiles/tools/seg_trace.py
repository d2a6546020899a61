#!/usr/bin/env python
"""Phase breakdown of k_seg_fb from a -DHF_SEG_TRACE build (HF_SEG_TRACE_FILE=<file>): per-workgroup s_memtime stamps of thread 0."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.int64)
names = ["fill_tab", "slow_base", "lane product", "scans", "barrier", "carry", "fwd replay", "bwd replay", "barrier2", "tail"]
d = np.diff(t[:, :11], axis=1)
n = t[:, 11]
print("workgroups", len(t), "windows per segment min/med/max", n.min(), int(np.median(n)), n.max())
for k, nm in enumerate(names):
    print(f"{nm:14s} median {np.median(d[:, k]):9.0f}  p90 {np.percentile(d[:, k], 90):9.0f}  cycles")
tot = t[:, 10] - t[:, 0]
print(f"{'total':14s} median {np.median(tot):9.0f}  p90 {np.percentile(tot, 90):9.0f}  max {tot.max()}")
rt = t[:, 12]
print("kernel span (100 MHz realtime, end stamps): %.1f us" % ((rt.max() - rt.min()) / 100.0))
order = np.argsort(rt)
print("end-time quantiles (us since first end):", [round((np.percentile(rt, q) - rt.min()) / 100.0, 1) for q in (10, 50, 90, 100)])
