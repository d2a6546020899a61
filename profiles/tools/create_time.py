#!/usr/bin/env python
"""Time of hf_create (upload, window records, tiles, statistics plan) on BASELINE configs[2]; the first call includes HIP start-up."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flagger_amd import hmm, synth
store = synth.config(2)
K = hmm.getBestNumberOfCollapsedComps(store)
model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
for i in range(3):
    t0 = time.perf_counter(); em = hmm.EMList(store, model); t1 = time.perf_counter()
    print("hf_create %.1f ms" % ((t1 - t0) * 1e3)); em.close()
os.environ["HF_STATS"] = "chunks"
t0 = time.perf_counter(); em = hmm.EMList(store, model); t1 = time.perf_counter()
print("hf_create (HF_STATS=chunks: plan still built) %.1f ms" % ((t1 - t0) * 1e3)); em.close()
