#!/usr/bin/env python
"""Per-kernel average time (us) of the E-step on BASELINE configs[2] with FIXED parameters (no M-step, errors ignored):
for timing builds whose results are deliberately wrong (-DHF_PROBE_* switches that remove one cost at a time)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from flagger_amd import hmm, synth  # noqa: E402
from flagger_amd import _native as N  # noqa: E402

store = synth.config(int(os.environ.get("PROBE_CONFIG", "2")))
K = hmm.getBestNumberOfCollapsedComps(store)
model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
em = hmm.EMList(store, model)
em.set_profiling(True)
for _ in range(30):
    try:
        em.em_iterate(model, False, 1e-3, mode=int(os.environ.get("PROBE_MODE", "0")))   # 1: forward-only passes
    except N.HFError:
        pass
ks = {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in em.kernel_time_sums().items() if v[1]}
print(sys.argv[1] if len(sys.argv) > 1 else "", ks, "sum", round(sum(ks.values()), 1))
em.close()
