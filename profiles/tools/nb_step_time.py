#!/usr/bin/env python
"""ms per EM step of the negative_binomial model on BASELINE configs[2] (informational; the headline is bench.py)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from flagger_amd import hmm, synth  # noqa: E402

store = synth.config(2)
K = hmm.getBestNumberOfCollapsedComps(store)
for mt, name in ((hmm.MODEL_NEGATIVE_BINOMIAL, "negative_binomial"), (hmm.MODEL_GAUSSIAN, "gaussian")):
    model = hmm.createModel(mt, K, store, np.zeros((4, 4)))
    em = hmm.EMList(store, model)
    n = 20
    for _ in range(50):
        em.em_iterate(model, True, 1e-3)
    t0 = time.perf_counter()
    for _ in range(n):
        em.em_iterate(model, True, 1e-3)
    dt = (time.perf_counter() - t0) / n          # no events in these steps
    em.set_profiling(True)                       # ... and every kernel bracketed for the breakdown
    for _ in range(n):
        em.em_iterate(model, True, 1e-3)
    ks = {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in em.kernel_time_sums().items() if v[1]}
    print(f"{name}: {dt * 1e3:.3f} ms/step, {store.n_windows / dt / 1e9:.2f} G windows/s, LL {model.loglikelihood:.1f}, kernels us {ks}")
    em.close()
