"""How close do realistic inputs come to a label flip?  (the segment kernels compute f·(T∘e) with fused
multiply-adds where the reference computes (f·T)·e, hmm.c:407, so a label can differ from the reference's where the two
largest posteriors are equal to the last few ulps — tests/test_neartie_gpu.py constructs such inputs.)

Here the question is asked of the inputs a user has: BASELINE configs[2] and configs[4] at full size and the reference simulator's three
100 000-observation tracks, each AFTER the EM has converged (the pass that writes the BED, hmm_flagger.c:464).  Both sides
decode with the SAME converged parameters; for every window the relative gap between the two largest ORACLE posteriors
is binned (< 1e-12, < 1e-10, < 1e-8) and the HIP <-> oracle label mismatches are counted per bin.  Asserted: no mismatch
outside the 1e-12 band.  The counts go to gpurun_out/label_margin.json (DESIGN.md §2 quotes the committed copy)."""
import json
import os

import numpy as np
import pytest

from flagger_amd import _native as N
from flagger_amd import hmm, synth
from flagger_amd.io import Table
from oracle_py import Oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

CASES = [("configs[2], full size", None, hmm.MODEL_TRUNC_EXP_GAUSSIAN, 1e-3),
         # the 7-region ONT-R10 workload (region-change windows with the uniform T = 0.2 of hmm.c:398-400, K = 10) had not been studied
         ("configs[4], full size", "cfg4", hmm.MODEL_TRUNC_EXP_GAUSSIAN, 1e-3),
         # the reference's real defaults — `-x hifi` / `-x ont-r9` with no -W: 16 kb windows (hmm_flagger.c:27,44,951-953), ~380 k
         # windows for a human diploid assembly, alpha all zero (the preset arrays are `int`, :21,36), minReadFractionAtEnds 0.95 / 1.0
         ("hifi preset defaults (16 kb windows), genome size", "hifi16k", hmm.MODEL_TRUNC_EXP_GAUSSIAN, 1e-3),
         ("ont-r9 preset defaults (16 kb windows), genome size", "r9_16k", hmm.MODEL_TRUNC_EXP_GAUSSIAN, 1e-3),
         ("sim100k_exp_gaussian", "sim100k_exp_gaussian.cov.gz", hmm.MODEL_TRUNC_EXP_GAUSSIAN, 1e-4),
         ("sim100k_gaussian", "sim100k_gaussian.cov.gz", hmm.MODEL_GAUSSIAN, 1e-4),
         ("sim100k_negative_binomial", "sim100k_negative_binomial.cov.gz", hmm.MODEL_NEGATIVE_BINOMIAL, 1e-4)]


@pytest.mark.parametrize("name,cov,model_type,tol", CASES, ids=[c[0].split(",")[0].split(" (")[0] for c in CASES])
def test_margin_of_the_final_labels(name, cov, model_type, tol):
    frac = 0.95
    if cov is None:
        store, alpha, K, adjust = synth.config(2), synth.HIFI_ALPHA, None, True
        max_mapq, min_mapq = 0.25, 0.75
    elif cov == "cfg4":   # -x ont-r10: minReadFractionAtEnds 0.8 (hmm_flagger.c:36-58)
        store, alpha, K, adjust, frac = synth.config(4), synth.ONT_R10_ALPHA, None, True, 0.8
        max_mapq, min_mapq = 0.25, 0.75
    elif cov in ("hifi16k", "r9_16k"):
        store, alpha, K, adjust, frac = synth.config(7), np.zeros((4, 4)), None, True, (0.95 if cov == "hifi16k" else 1.0)
        assert store.window_len == 16000 and 350_000 < store.n_windows < 420_000
        max_mapq, min_mapq = 0.25, 0.75
    else:   # the docs/hmm_test recipe (tests/test_cli_gpu.py): --chunkLen 1000 --windowLen 1 --collapsedComps 4 --minHighMapqRatio 0 -e
        store, alpha, K, adjust = Table(os.path.join(GOLD, cov), 1000, 1).store(), np.zeros((4, 4)), 4, False
        max_mapq, min_mapq = 0.25, 0.0
    K = hmm.getBestNumberOfCollapsedComps(store) if K is None else K
    model = hmm.createModel(model_type, K, store, alpha, max_mapq, min_mapq)
    em = hmm.EMList(store, model, adjust, frac)
    orc = Oracle(store, model_type, K, alpha, max_mapq=max_mapq, min_mapq=min_mapq, adjust=adjust, min_read_frac=frac, threads=16)
    try:
        iters = 0
        for iters in range(1, 41 if model_type == hmm.MODEL_NEGATIVE_BINOMIAL else 101):     # hmm_flagger.c:337-445
            hmm.EM_runOneIterationForList(em, model)
            converged = hmm.HMM_estimateParameters(model, tol)
            hmm.HMM_resetEstimators(model)
            if converged:
                break
        orc.set_param_vector(model.param_vector())               # the same converged parameters on both sides
        hmm.EM_runOneIterationForList(em, model)                  # the final inference pass
        assert orc.run_iteration() == 0
        lab, olab = em.labels(), orc.labels()
        f, b, sc = orc.forward_backward()
        post = f * b * sc[:, None]
        post /= post.sum(axis=1, keepdims=True)
        srt = np.sort(post, axis=1)
        gap = (srt[:, 3] - srt[:, 2]) / srt[:, 3]
        mism = lab != olab
        rec = {"windows": int(lab.size), "em_iterations": iters, "label_mismatches": int(mism.sum()),
               "smallest_relative_gap_of_the_top_two_posteriors": float(gap.min())}
        for band in (1e-12, 1e-10, 1e-8, 1e-6):
            inside = gap < band
            rec["gap<%g" % band] = {"windows": int(inside.sum()), "mismatches": int((mism & inside).sum())}
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "label_margin.json")
        allrec = json.load(open(path)) if os.path.exists(path) else {}
        allrec[name] = rec
        json.dump(allrec, open(path, "w"), indent=1)
        assert not (mism & (gap >= 1e-12)).any(), rec
    finally:
        em.close()
        orc.close()
