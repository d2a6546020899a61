#!/usr/bin/env python
"""Soak test of the polled completion (hf_finish without stream synchronisation): alternate two parameter sets for N passes
per statistics mode and compare every returned vector with the one of its own parameters; prints what differs.
  python profiles/tools/poll_soak.py 400000 [nb]      (history: stamp only: 1 stale pass in ~2 000; + XOR checksum: 1 in ~300 000,
  two equal stale words cancel; + position-weighted checksum: 0 in 800 000)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, "tests")
import numpy as np
from flagger_amd import hmm, synth, _native as N
store = synth.config(2, scale=0.01)
K = 4
MT = hmm.MODEL_NEGATIVE_BINOMIAL if len(sys.argv) > 2 and sys.argv[2] == "nb" else hmm.MODEL_TRUNC_EXP_GAUSSIAN
ALPHA = np.zeros((4, 4)) if MT == hmm.MODEL_NEGATIVE_BINOMIAL else synth.HIFI_ALPHA
model_a = hmm.createModel(MT, K, store, ALPHA)
model_b = hmm.createModel(MT, K, store, ALPHA)
em = hmm.EMList(store, model_a)
hmm.EM_runOneIterationForList(em, model_b); hmm.HMM_estimateParameters(model_b, 1e-3)
for mode in (N.HF_STATS_ROWS, N.HF_STATS_CHUNKS):
    em.set_stats_mode(mode)
    em.launch(model_a); ref_a = em.finish().copy()
    em.launch(model_b); ref_b = em.finish().copy()
    nbad = 0
    for i in range(int(sys.argv[1])):
        m, ref, other = (model_a, ref_a, ref_b) if i % 2 == 0 else (model_b, ref_b, ref_a)
        em.launch(m)
        got = em.finish()
        if not np.array_equal(got, ref):
            d = np.where(got != ref)[0]
            print("mode", mode, "iter", i, "n_diff", d.size, "first", d[:6], "equals other pass there:", np.array_equal(got[d], other[d]),
                  "rel", np.max(np.abs(got[d] - ref[d]) / np.maximum(np.abs(ref[d]), 1e-300)))
            nbad += 1
            if nbad >= 5: break
    print("mode", mode, "bad", nbad)
