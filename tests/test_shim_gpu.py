"""integration/hmm_hip_shim.c end to end on a GPU: the reference-shaped objects of tests/shim_mock (EM list, HMM with its
estimator objects, Inference records) go through EM_runForwardForList / EM_runOneIterationForList / EM_getPosterior of the shim;
what lands in them must be what the C ABI returns directly: log-likelihood, every estimator numerator / denominator, the
transition counts, the predictions, the posteriors — bit for bit (same library, same context parameters)."""
import numpy as np
import pytest

from flagger_amd import _native as N
from flagger_amd import hmm, synth
from test_shim_cpu import build_driver, write_dump

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model_type,cfg", [(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 2), (hmm.MODEL_GAUSSIAN, 4), (hmm.MODEL_NEGATIVE_BINOMIAL, 2)])
def test_shim_scatters_what_the_abi_returns(model_type, cfg, tmp_path):
    import subprocess
    store = synth.config(cfg, scale=0.01)
    alpha = synth.HIFI_ALPHA if cfg == 2 else synth.ONT_R10_ALPHA
    K = min(hmm.getBestNumberOfCollapsedComps(store), 6)
    model = hmm.createModel(model_type, K, store, alpha)
    exe = build_driver(tmp_path)
    dump, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_dump(dump, store, model, True, 0.9)
    r = subprocess.run([exe, dump, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    em = hmm.EMList(store, model, True, 0.9)
    try:
        hmm.EM_runForwardForList(em, model)
        ll_fwd = model.loglikelihood
        hmm.EM_runOneIterationForList(em, model)
        ref = model.estimators.copy()
        lab = em.labels()
        R = model.numberOfRegions
        raw = open(out, "rb").read()
        n_stats = R * (24 * K + 16)
        head = np.frombuffer(raw, dtype="<f8", count=2 + n_stats)
        assert head[0] == ll_fwd and head[1] == ref[0]
        assert np.array_equal(head[2:], ref[1:])
        o = (2 + n_stats) * 8
        assert np.array_equal(np.frombuffer(raw, dtype=np.int8, count=store.n_windows, offset=o), lab)
        post = np.frombuffer(raw, dtype="<f8", count=8, offset=o + store.n_windows).reshape(2, 4)
        assert np.array_equal(post[0], em.posterior(0, 1)[0]) and np.array_equal(post[1], em.posterior(store.n_windows - 1, 1)[0])
    finally:
        em.close()
