#!/usr/bin/env python
"""k_seg_fb<true> from a -DHF_SEG_TRACE build (profiles/tools/seg_trace.sh; HF_SEG_TRACE_FILE=<file>): s_memtime stamps of the
phases of every workgroup (one wavefront), the per-step laps of both replays, workgroup lifetimes (s_memrealtime) and how
the workgroups spread over the CUs (HW_ID)."""
import sys
import numpy as np
N = 25
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, N).astype(np.int64)
t = t[t[:, 19] > 0]
L = t[:, 19].astype(float)
print("workgroups", len(t), " windows per lane: mean %.2f" % L.mean(), " windows per segment: mean %.0f" % t[:, 20].mean())
names = ["row indices, offset table, lane product (8 row fetches)", "park, prefix scan, product published", "unpark + suffix scan (before the other segments are awaited)",
         "wait for the other segments + both chains", "carried-in f", "direction of b", "forward replay",
         "log-likelihood", "b at the last window, label", "backward replay", "last record, labels out"]
life = (t[:, 11] - t[:, 0]).astype(float)
print("cycles per workgroup (mean), total %.0f:" % life.mean())
for k, nm in enumerate(names):
    d = (t[:, k + 1] - t[:, k]).astype(float)
    print("  %-58s %8.0f  %5.1f %%" % (nm, d.mean(), 100 * d.mean() / life.mean()))
laps = ["fwd: wait for the DMA + read the rows out of LDS", "fwd: issue the next step's row fetch", "fwd: arithmetic",
        "bwd: wait + read", "bwd: record through LDS + stores", "bwd: issue", "bwd: arithmetic + label"]
for k, nm in enumerate(laps):
    per = t[:, 12 + k] / np.maximum(L - (1 if k >= 3 else 0), 1)
    print("  per step, %-48s %6.0f cycles" % (nm, per.mean()))
hw, xcc = t[:, 22], t[:, 23]
key = (xcc & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 20 + ((hw >> 8) & 0xf)
u, c = np.unique(key, return_counts=True)
print("CUs used", len(u), " workgroups per CU: histogram", {int(k): int(v) for k, v in enumerate(np.bincount(c)) if v})
e = t[:, 21]
print("first workgroup starts .. last ends: %.0f cycles (s_memtime)" % float(t[:, 11].max() - t[:, 0].min()))
print("workgroup end (s_memrealtime, us after the first end): p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" %
      tuple((np.percentile(e, q) - e.min()) / 100.0 for q in (10, 50, 90, 100)))
print("lifetime cycles: p10 %.0f  p50 %.0f  p90 %.0f  max %.0f" % tuple(np.percentile(life, q) for q in (10, 50, 90, 100)))
# who is slow: the phases of the slowest and the fastest tenth of the workgroups, lifetimes by XCD, by SIMD slot, by windows per lane
order = np.argsort(life)
n10 = max(1, len(order) // 10)
fast, slow = order[:n10], order[-n10:]
print("phase means, fastest tenth | slowest tenth of the workgroups (by lifetime):")
for k, nm in enumerate(names):
    d = (t[:, k + 1] - t[:, k]).astype(float)
    print("  %-58s %8.0f | %8.0f" % (nm, d[fast].mean(), d[slow].mean()))
print("start (cycles after the first start): fastest tenth %.0f, slowest tenth %.0f" % ((t[fast, 0] - t[:, 0].min()).mean(), (t[slow, 0] - t[:, 0].min()).mean()))
print("windows per lane: fastest tenth %.2f, slowest tenth %.2f" % (L[fast].mean(), L[slow].mean()))
for x in sorted(set((xcc & 0xf).tolist())):
    sel = (xcc & 0xf) == x
    print("  XCD %d: %4d workgroups, lifetime mean %.0f  p90 %.0f" % (x, int(sel.sum()), life[sel].mean(), np.percentile(life[sel], 90)))
# dispatch skew (s_memrealtime, 100 MHz, one clock for the device): starts and ends relative to the launch's first start; the
# trace buffer is shared by every context of the process, so only workgroups of the LAST launch are looked at
ws, we = t[:, 24].astype(float) / 100.0, t[:, 21].astype(float) / 100.0
last = we > we.max() - 300.0
print("workgroups of the last launch: %d of %d" % (int(last.sum()), len(t)))
ws0 = ws[last].min()
print("start, us after the launch's first start: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(ws[last] - ws0, q) for q in (10, 50, 90, 100)))
print("end,   us after the launch's first start: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % tuple(np.percentile(we[last] - ws0, q) for q in (10, 50, 90, 100)))
print("life (wall), us: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % tuple(np.percentile((we - ws)[last], q) for q in (10, 50, 90, 100)))
print("correlation(start, life) %.2f   correlation(start, end) %.2f" % (np.corrcoef(ws[last], (we - ws)[last])[0, 1], np.corrcoef(ws[last], we[last])[0, 1]))
# by position of the segment inside its chunk's run of workgroups (blockIdx order): first / middle / last thirds of the grid
gi = np.arange(len(t))[last]
for nm, sel in (("first third of the grid", gi < len(t) / 3), ("middle third", (gi >= len(t) / 3) & (gi < 2 * len(t) / 3)), ("last third", gi >= 2 * len(t) / 3)):
    print("  %-24s start mean %.1f us, life mean %.1f us, end mean %.1f us" % (nm, (ws[last][sel] - ws0).mean(), (we - ws)[last][sel].mean(), (we[last][sel] - ws0).mean()))
simd = (hw >> 4) & 3
for x in range(4):
    sel = simd == x
    if sel.any(): print("  SIMD %d: %4d workgroups, lifetime mean %.0f" % (x, int(sel.sum()), life[sel].mean()))
