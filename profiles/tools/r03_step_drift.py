import sys, time
sys.path.insert(0, ".")
import torch
from flagger_amd import hmm, synth, _native as N
store = synth.config(2)
K = hmm.getBestNumberOfCollapsedComps(store)
model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
em = hmm.EMList(store, model)
def step():
    if hasattr(em, "em_iterate"): em.em_iterate(model, 1e-3)
    else:
        hmm.EM_runOneIterationForList(em, model); hmm.HMM_estimateParameters(model, 1e-3); hmm.HMM_resetEstimators(model)
for _ in range(3): step()
out = []
for blk in range(60):
    t0 = time.perf_counter()
    for _ in range(100): step()
    out.append((time.perf_counter() - t0) / 100 * 1e3)
print(" ".join("%.4f" % x for x in out))
print("ll", model.loglikelihood)
# where exactly: per-step times of a fresh context
em.close()
for trial in range(2):
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
    em = hmm.EMList(store, model)
    ts = []
    for i in range(1500):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    big = [(i, round(t * 1e3, 2)) for i, t in enumerate(ts) if t > 1e-3]
    print("trial", trial, "steps slower than 1 ms:", big)
    em.close()
