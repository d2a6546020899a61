#!/usr/bin/env python
"""hf_create on BASELINE configs[2] the way bench.py's em_run leg meets it: a first context that has run passes stays ALIVE, then
fresh contexts are created one after another (each runs two passes and is closed).  Prints every context's wall time and phases
(hf_create_phases) — what a box pays that the third-context figure of profiles/r05_hf_create.txt hides."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from flagger_amd import hmm, synth
store = synth.config(2)
K = hmm.getBestNumberOfCollapsedComps(store)
model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
t0 = time.perf_counter(); em0 = hmm.EMList(store, model); t1 = time.perf_counter()
print("first context of the process: %.1f ms" % ((t1 - t0) * 1e3))
for _ in range(int(os.environ.get("PROBE_PASSES", "25"))):
    em0.em_iterate(model, True, 1e-3)
idle = float(os.environ.get("PROBE_IDLE_S", "0"))
tot = []
for i in range(int(os.environ.get("PROBE_N", "9"))):
    if idle:
        time.sleep(idle)
    m = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
    t0 = time.perf_counter(); em = hmm.EMList(store, m); t1 = time.perf_counter()
    ph = em.create_phases()
    tot.append((t1 - t0) * 1e3)
    print("context %d: wall %.2f ms (library %.2f)  " % (i, tot[-1], ph.get("total", 0)) +
          " | ".join("%s %.2f" % (k[:28], v) for k, v in ph.items() if v >= 0.05 and k != "total"))
    em.em_iterate(m, True, 1e-3); em.em_iterate(m, True, 1e-3)
    em.close()
print("median %.2f  min %.2f  max %.2f ms over %d fresh contexts (first context alive)" % (np.median(tot), min(tot), max(tot), len(tot)))
