"""GPU parity: the HIP E-step (through the C ABI) against the oracle on the same seeded inputs.

Bar (BASELINE.json north_star): labels bit-exact; log-likelihood within 1e-6 relative (we hold
1e-9); the fp64 sufficient statistics / forward / backward within 1e-9 relative — the only
permitted difference is the last-ulp behaviour of exp()/log() on the device vs glibc.
"""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from flagger_amd import _native as N
from flagger_amd import hmm, synth
from oracle_py import Oracle

pytestmark = pytest.mark.gpu

LL_RTOL = 1e-9      # north_star asks for 1e-6
STAT_RTOL = 1e-9


SCAN_CHUNKS = 100 + N.HF_ALGO_SCAN


def make_em(store, model, *args, algo=N.HF_ALGO_SCAN, **kw):
    """EMList for an ALGOS parameter."""
    em = hmm.EMList(store, model, *args, algo=algo % 100, **kw)
    if algo >= 100:
        em.set_stats_mode(N.HF_STATS_CHUNKS)
        assert em.stats_mode == N.HF_STATS_CHUNKS
    return em


def _check_pass(store, model_type, K, alpha, algo, adjust=True, frac=0.95, n_iter=2, max_mapq=0.25, min_mapq=0.75):
    model = hmm.createModel(model_type, K, store, alpha, max_mapq, min_mapq)
    em = make_em(store, model, adjust, frac, device=0, algo=algo)
    orc = Oracle(store, model_type, K, alpha, max_mapq, min_mapq, adjust, frac, threads=8)
    try:
        assert np.array_equal(model.param_vector(), orc.param_vector())
        for _ in range(n_iter):
            hmm.EM_runOneIterationForList(em, model)
            assert orc.run_iteration() == 0
            ref = orc.stats_vector(model.maxNumberOfComps)
            got = model.estimators
            assert abs(got[0] - ref[0]) <= LL_RTOL * abs(ref[0]), (got[0], ref[0])
            scale = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
            assert np.all(np.abs(got - ref) <= STAT_RTOL * scale), np.max(np.abs(got - ref) / scale)
            lab, olab = em.labels(), orc.labels()
            assert np.array_equal(lab, olab), f"{np.count_nonzero(lab != olab)} label mismatches of {lab.size}"
            f, b, sc = em.forward_backward()
            of, ob, osc = orc.forward_backward()
            assert np.allclose(sc, osc, rtol=1e-10, atol=0)
            assert np.allclose(f, of, rtol=1e-9, atol=1e-300)
            assert np.allclose(b, ob, rtol=1e-9, atol=1e-300)
            c1 = hmm.HMM_estimateParameters(model, 1e-3)
            hmm.HMM_resetEstimators(model)
            c2 = orc.estimate_parameters(1e-3)
            assert c1 == c2
            assert np.allclose(model.param_vector(), orc.param_vector(), rtol=1e-8, atol=1e-300)
            # continue from the ORACLE's parameters so both sides see identical inputs next pass
            model.set_param_vector(orc.param_vector())
    finally:
        em.close()
        orc.close()


# "scan" takes the library default (statistics by emission row where it applies), "scan-chunks" the per-chunk vectors
ALGOS = [pytest.param(N.HF_ALGO_SEQ, id="seq"), pytest.param(N.HF_ALGO_SCAN, id="scan"),
         pytest.param(SCAN_CHUNKS, id="scan-chunks")]


@pytest.mark.parametrize("algo", ALGOS)
def test_cfg1_fixed_parameter_decode(algo):
    """BASELINE configs[1]: 1 contig 10 Mb, 4 kb windows, fixed parameters (--iterations 0)."""
    store = synth.config(1)
    assert store.n_windows == 2500 and store.n_chunks == 1
    _check_pass(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, hmm.getBestNumberOfCollapsedComps(store), synth.HIFI_ALPHA,
                algo, n_iter=1)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("model_type", [hmm.MODEL_TRUNC_EXP_GAUSSIAN, hmm.MODEL_GAUSSIAN])
@pytest.mark.parametrize("alpha_name", ["zero", "hifi", "ont"])
def test_small_diploid(algo, model_type, alpha_name):
    alpha = {"zero": np.zeros((4, 4)), "hifi": synth.HIFI_ALPHA, "ont": synth.ONT_R10_ALPHA}[alpha_name]
    store = synth.config(2, scale=0.01)
    K = hmm.getBestNumberOfCollapsedComps(store)
    _check_pass(store, model_type, K, alpha, algo)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("K", [2, 6, 16])
def test_negative_binomial_model(algo, K):
    """--modelType negative_binomial: emission depends on x only, statistics through per-state count
    data and the digamma table; theta bound across all states, lambda bound with the mean factors."""
    store = synth.config(2, scale=0.004)
    _check_pass(store, hmm.MODEL_NEGATIVE_BINOMIAL, K, np.zeros((4, 4)), algo, n_iter=3)


@pytest.mark.parametrize("algo", ALGOS)
def test_negative_binomial_regions_ragged_chunks_and_extreme_coverage(algo):
    store = synth.synthesize([300_000, 5_000, 9_000, 64_000, 65_000, 1_000], 1000, 40_000, [20, 30, 12], seed=77,
                             region_run_bases=(5_000, 60_000))
    store.cov[::97] = 250          # folded into count-data bin 249 (count_data.c:49-57)
    store.cov[5::89] = 0
    store.mapq[:] = store.cov      # keep the collapsed state valid where coverage is high
    _check_pass(store, hmm.MODEL_NEGATIVE_BINOMIAL, 5, synth.HIFI_ALPHA, algo, n_iter=2, min_mapq=0.0)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("K", [2, 5, 10, 16])
def test_collapsed_components(algo, K):
    store = synth.config(2, scale=0.004)
    _check_pass(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, synth.HIFI_ALPHA, algo)


@pytest.mark.parametrize("algo", ALGOS)
def test_multi_region_ont(algo):
    """configs[4] shape: 7 bias regions, 8 kb windows, region changes inside chunks."""
    store = synth.config(4, scale=0.01)
    assert store.n_regions == 7 and len(np.unique(store.regions())) > 3
    _check_pass(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, hmm.getBestNumberOfCollapsedComps(store), synth.ONT_R10_ALPHA,
                algo, frac=0.8)


@pytest.mark.parametrize("algo", ALGOS)
def test_wide_coverage_support_worst_case_for_the_tables(algo):
    """synth.config(5): coverage spread over 0..250 in 7 regions, K = 10 — nearly every window has its own emission row: the
    statistics plan is the compact one (groups back to back, several batches of row slots per wavefront) and the row tables leave the L2."""
    store = synth.config(5, scale=0.01)
    K = hmm.getBestNumberOfCollapsedComps(store)
    assert K == 10 and store.n_regions == 7 and int(store.cov.max()) == 250
    _check_pass(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, synth.ONT_R10_ALPHA, algo, n_iter=2)


def test_over_dispersed_coverage():
    """synth.config(6): configs[2] with negative-binomial coverage (variance = 3 x mean, a heavy right tail): more emission keys
    and more collapsed components than the Gaussian workload, still on the statistics-by-row path; all three model types."""
    store = synth.config(6, scale=0.02)
    K = hmm.getBestNumberOfCollapsedComps(store)
    assert int(store.cov.max()) > 100
    for mt, alpha in ((hmm.MODEL_TRUNC_EXP_GAUSSIAN, synth.HIFI_ALPHA), (hmm.MODEL_GAUSSIAN, synth.HIFI_ALPHA),
                      (hmm.MODEL_NEGATIVE_BINOMIAL, np.zeros((4, 4)))):
        _check_pass(store, mt, min(K, 6), alpha, N.HF_ALGO_SCAN, n_iter=2)


@pytest.mark.parametrize("algo", ALGOS)
def test_empty_chunk_list(algo):
    """No chunk at all (e.g. --contigsList that matches nothing): log-likelihood 0, all-zero statistics, no labels."""
    full = synth.synthesize([50_000], 1000, 20_000, [20], seed=2)
    store = full.subset_chunks([])
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 3, full, synth.HIFI_ALPHA)
    em = make_em(store, model, algo=algo)
    hmm.EM_runOneIterationForList(em, model)
    assert model.loglikelihood == 0.0 and not np.any(model.estimators)
    assert em.labels().size == 0
    hmm.EM_runForwardForList(em, model)
    assert model.loglikelihood == 0.0
    em.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_one_long_chunk_spanning_many_tiles(algo):
    """A 45 000-window chunk = 88 segments of the segment kernels (hf_seg.h): more products of OTHER segments than k_seg_fb
    stages in LDS (HF_SEG_PSTAGE = 24; the rest are read from global memory), next to a chunk of one half-filled segment
    and one of exactly 64 full segments."""
    store = synth.synthesize([45_000_000, 256_000, 32_768_000], 1000, 60_000_000, [20], seed=21)
    assert sorted(np.diff(store.chunk_off).tolist()) == [256, 32768, 45000]
    _check_pass(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, 4, synth.HIFI_ALPHA, algo, n_iter=1)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("model_type", [hmm.MODEL_TRUNC_EXP_GAUSSIAN, hmm.MODEL_NEGATIVE_BINOMIAL])
def test_maximum_number_of_regions_and_components(algo, model_type):
    """All 64 region classes the 6 region bits allow (ptBlock.c:294-304) with 16 collapsed components: the per-region
    tables no longer fit the default LDS budget of the tile kernels."""
    store = synth.synthesize([600_000, 90_000], 1000, 100_000, [20 + (i % 7) for i in range(64)], seed=5,
                             region_run_bases=(3_000, 20_000))
    assert len(np.unique(store.annot >> np.uint64(58))) > 12 and int((store.annot >> np.uint64(58)).max()) > 60
    _check_pass(store, model_type, 16, synth.HIFI_ALPHA, algo, n_iter=1)


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("adjust,avg_len", [(False, 15000), (True, 0), (True, 15000), (True, 200000)])
def test_contig_end_adjustment(algo, adjust, avg_len):
    """beta: disabled (-e), missing #avg_alignment_len (=> 0.25 everywhere, Q4), normal, reads longer than chunks."""
    store = synth.synthesize([700_000, 90_000, 4_100], 1000, 200_000, [20], seed=11, avg_alignment_len=avg_len)
    _check_pass(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, 4, synth.HIFI_ALPHA, algo, adjust=adjust)


@pytest.mark.parametrize("algo", ALGOS)
def test_ragged_and_tiny_chunks(algo):
    """chunks of 1, 2, 3, 63, 64, 65, 129 ... windows (tile edges of the kernels) in one list."""
    W = 100
    lens = [1 * W, 2 * W, 3 * W, 63 * W, 64 * W, 65 * W, 129 * W, 1000 * W + 37, 5 * W - 1, 2049 * W]
    store = synth.synthesize(lens, W, 10_000_000, [20, 25], seed=5, region_run_bases=(2_000, 30_000))
    assert sorted(np.diff(store.chunk_off))[:3] == [1, 2, 3]
    _check_pass(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, 3, synth.HIFI_ALPHA, algo)


@pytest.mark.parametrize("algo", ALGOS)
def test_extreme_coverage_values(algo):
    """coverage 0 and 250 (clip value), Dup/Col validity masks on and off, clip-driven End column."""
    rng = np.random.default_rng(3)
    store = synth.synthesize([2_000_000], 1000, 500_000, [20], seed=9)
    n = store.n_windows
    store.cov[rng.integers(0, n, 200)] = 250
    store.cov[rng.integers(0, n, 200)] = 0
    store.mapq[:] = (store.cov * rng.uniform(0, 1, n)).astype(np.uint16)
    store.clip[:] = (store.cov * rng.uniform(0, 1.3, n)).astype(np.uint16)
    _check_pass(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, 6, synth.HIFI_ALPHA, algo)
    _check_pass(store, hmm.MODEL_GAUSSIAN, 6, np.zeros((4, 4)), algo, max_mapq=0.5, min_mapq=0.5)


@pytest.mark.parametrize("algo", ALGOS)
def test_forward_only_mode(algo):
    """EM_runForwardForList (SQUAREM line search): log-likelihood only."""
    store = synth.config(2, scale=0.004)
    K = 4
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
    em = make_em(store, model, algo=algo)
    orc = Oracle(store, 0, K, synth.HIFI_ALPHA)
    hmm.EM_runForwardForList(em, model)
    assert orc.run_iteration(forward_only=True) == 0
    ref = orc.m.contents.loglikelihood
    assert abs(model.loglikelihood - ref) <= LL_RTOL * abs(ref)
    em.close()
    orc.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_full_em_against_oracle(algo, tmp_path):
    """Whole EM run (runHMMFlagger): every row of loglikelihood.tsv within 1e-6 relative, final labels
    identical, final parameters equal to print precision."""
    store = synth.config(2, scale=0.01)
    K = hmm.getBestNumberOfCollapsedComps(store)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
    em = make_em(store, model, algo=algo)
    out = tmp_path / "gpu"
    out.mkdir()
    lls = hmm.runHMMFlagger(em, model, 12, 1e-3, str(out))
    orc = Oracle(store, 0, K, synth.HIFI_ALPHA)
    oout = tmp_path / "oracle"
    oout.mkdir()
    olls = orc.run_em(12, 1e-3, str(oout))
    assert len(lls) == len(olls)
    assert np.allclose(lls, olls, rtol=1e-6, atol=0)
    assert np.array_equal(em.labels(), orc.labels())
    assert np.allclose(model.param_vector(), orc.param_vector(), rtol=1e-6, atol=1e-300)
    for name in ("loglikelihood.tsv", "emission_final.tsv", "transition_final.tsv", "emission_initial.tsv"):
        assert (out / name).read_text() == (oout / name).read_text(), name
    em.close()
    orc.close()


@pytest.mark.parametrize("model_type", [hmm.MODEL_TRUNC_EXP_GAUSSIAN, hmm.MODEL_NEGATIVE_BINOMIAL])
def test_forward_only_loglikelihood_equals_the_full_pass_bit_for_bit(model_type):
    """SQUAREM's line search (hmm.c:904-914) ends when EM_runForwardForList on a copy of model 0 returns model 0's
    log-likelihood from EM_runOneIterationForList: the reference — and the re-pointed reference (integration/hmm_hip_shim.c) —
    relies on the two passes summing alike.  Both statistics paths, both algorithms, several shapes."""
    for seed, scale in ((1, 0.004), (2, 0.013), (3, 0.05)):
        store = synth.config(2 if seed < 3 else 4, scale=scale)
        K = 4
        alpha = np.zeros((4, 4)) if model_type == hmm.MODEL_NEGATIVE_BINOMIAL else synth.HIFI_ALPHA
        model = hmm.createModel(model_type, K, store, alpha)
        for algo in (N.HF_ALGO_SEQ, N.HF_ALGO_SCAN):
            for mode in ((N.HF_STATS_CHUNKS,) if algo == N.HF_ALGO_SEQ else (N.HF_STATS_ROWS, N.HF_STATS_CHUNKS)):
                em = make_em(store, model, algo=algo)
                em.set_stats_mode(mode)
                hmm.EM_runOneIterationForList(em, model)
                full = model.loglikelihood
                hmm.HMM_resetEstimators(model)
                hmm.EM_runForwardForList(em, model)
                assert model.loglikelihood == full, (seed, algo, mode)
                em.close()


def test_scale_underflow_is_reported():
    """The reference exits with 'scale ... is very low!' (hmm.c:412-415); the ABI returns HF_E_SCALE.
    Emissions are floored at 1e-40, so the scale only underflows through the transition matrix: put almost all
    mass on Err and feed a window above the trunc point (Err emission exactly 0)."""
    store = synth.synthesize([400_000], 1000, 1_000_000, [20], seed=2)
    store.cov[100:104] = 250
    for algo in (N.HF_ALGO_SEQ, N.HF_ALGO_SCAN):
        model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 2, store, np.zeros((4, 4)))
        v = model.param_vector().reshape(1, -1)
        t = v[0, :25].reshape(5, 5)
        t[:4, :4] = 1e-30
        t[:4, 0] = 1.0 - 1e-4
        model.set_param_vector(v.ravel())
        em = make_em(store, model, algo=algo)
        orc = Oracle(store, 0, 2, np.zeros((4, 4)))
        orc.set_param_vector(model.param_vector())
        assert orc.run_iteration() == -1
        with pytest.raises(N.HFError) as ei:
            hmm.EM_runOneIterationForList(em, model)
        assert ei.value.code == N.HF_E_SCALE
        em.close()
        orc.close()


def test_nan_emission_is_reported():
    """The reference exits with '[Error] prob is NAN' (hmm_utils.c:782-786); the ABI returns HF_E_NAN.  A negative variance
    makes sqrt(var*2*PI) NaN for every window; full and forward-only passes, both statistics paths, both algorithms."""
    store = synth.synthesize([400_000, 150_000], 1000, 200_000, [20], seed=3)
    K = 3
    for algo in (N.HF_ALGO_SEQ, N.HF_ALGO_SCAN):
        model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
        v = model.param_vector()
        km = (v.size - 27) // 12                 # [trans 25 | lambda | trunc | mean 4*KM | var 4*KM | weight 4*KM]
        v[27 + 4 * km + 2 * km] = -1.0           # variance of the Hap state
        model.set_param_vector(v)
        orc = Oracle(store, 0, K, synth.HIFI_ALPHA)
        orc.set_param_vector(model.param_vector())
        assert orc.run_iteration() < 0
        orc.close()
        for mode in ((N.HF_STATS_CHUNKS,) if algo == N.HF_ALGO_SEQ else (N.HF_STATS_ROWS, N.HF_STATS_CHUNKS)):
            em = make_em(store, model, algo=algo)
            em.set_stats_mode(mode)
            with pytest.raises(N.HFError) as ei:
                hmm.EM_runOneIterationForList(em, model)
            assert ei.value.code == N.HF_E_NAN
            with pytest.raises(N.HFError) as ei:
                hmm.EM_runForwardForList(em, model)
            assert ei.value.code == N.HF_E_NAN
            em.close()


def test_full_size_cfg2_one_pass_and_invariants():
    """BASELINE configs[2] at full size (1.5 M windows, 286 chunks): one E-pass against the oracle plus
    size-independent properties of the scaled forward-backward."""
    store = synth.config(2)
    assert 1_400_000 < store.n_windows < 1_700_000 and 250 < store.n_chunks < 330
    K = hmm.getBestNumberOfCollapsedComps(store)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
    em = hmm.EMList(store, model)
    # one M-step first so the pass runs with non-trivial parameters
    hmm.EM_runOneIterationForList(em, model)
    hmm.HMM_estimateParameters(model, 1e-3)
    hmm.EM_runOneIterationForList(em, model)
    got = model.estimators.copy()
    lab = em.labels()
    orc = Oracle(store, 0, K, synth.HIFI_ALPHA, threads=16)
    orc.set_param_vector(model.param_vector())
    assert orc.run_iteration() == 0
    ref = orc.stats_vector(K)
    assert abs(got[0] - ref[0]) <= LL_RTOL * abs(ref[0])
    scale = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
    assert np.all(np.abs(got - ref) <= STAT_RTOL * scale)
    olab = orc.labels()
    assert np.array_equal(lab, olab), f"{np.count_nonzero(lab != olab)} label mismatches of {lab.size}"
    # invariants: every pair (i, i+1), i = 1..T-2 contributes total xi mass 1 (after / terminationProb)
    T = np.diff(store.chunk_off)
    pairs = np.maximum(T - 2, 0).sum()
    base, st = 1, N.region_stride(K)
    trans = got[base + 24 * K: base + 24 * K + 16]
    assert abs(trans.sum() - pairs) <= 1e-7 * pairs
    # estimator identities of the reference: mean.den == var.den == weight.num; weight.den equal for all comps
    for s in (1, 2, 3):
        nc = K if s == 3 else 1
        md = got[base + ((s * 3 + 0) * 2 + 1) * K: base + ((s * 3 + 0) * 2 + 1) * K + nc]
        vd = got[base + ((s * 3 + 1) * 2 + 1) * K: base + ((s * 3 + 1) * 2 + 1) * K + nc]
        wn = got[base + ((s * 3 + 2) * 2 + 0) * K: base + ((s * 3 + 2) * 2 + 0) * K + nc]
        wd = got[base + ((s * 3 + 2) * 2 + 1) * K: base + ((s * 3 + 2) * 2 + 1) * K + nc]
        assert np.array_equal(md, vd) and np.array_equal(md, wn) and np.all(wd == wd[0])
        assert abs(wd[0] - md.sum()) <= 1e-9 * wd[0]
        assert abs(md.sum() - trans.reshape(4, 4)[:, s].sum()) <= 1e-9 * md.sum()
    post = em.posterior(0, 200_000)
    assert np.allclose(post.sum(axis=1), 1.0, atol=1e-12) and np.array_equal(post.argmax(axis=1), lab[:200_000])
    f, b, sc = em.forward_backward(0, 200_000)
    assert np.allclose(f.sum(axis=1), 1.0, atol=1e-12)
    assert np.allclose((f * b).sum(axis=1) * sc, 1e-4, rtol=1e-9)     # the scaling invariant used by the scan
    em.close()
    orc.close()


@pytest.mark.parametrize("model_type,cfg", [(hmm.MODEL_GAUSSIAN, 2), (hmm.MODEL_NEGATIVE_BINOMIAL, 2), (hmm.MODEL_TRUNC_EXP_GAUSSIAN, 6)],
                         ids=["gaussian", "negative_binomial", "over-dispersed coverage"])
def test_full_size_other_models_one_pass(model_type, cfg):
    """BASELINE configs[2]'s geometry at FULL size (1.5 M windows, 286 chunks) for the model types that are not the headline — `gaussian`,
    `negative_binomial` (hmm_utils.c:335-639: count data, digamma table) — and the headline model on over-dispersed coverage
    (synth.config(6): negative-binomial coverage, variance = 3 x mean, K = 9): one pass with non-trivial parameters against the
    oracle: labels identical, log-likelihood and statistics 1e-9."""
    store = synth.config(cfg)
    assert store.n_windows > 1_400_000
    K = hmm.getBestNumberOfCollapsedComps(store)
    alpha = np.zeros((4, 4)) if model_type == hmm.MODEL_NEGATIVE_BINOMIAL else synth.HIFI_ALPHA
    model = hmm.createModel(model_type, K, store, alpha)
    em = hmm.EMList(store, model)
    orc = Oracle(store, model_type, K, alpha, threads=16)
    try:
        hmm.EM_runOneIterationForList(em, model)
        hmm.HMM_estimateParameters(model, 1e-3)
        hmm.HMM_resetEstimators(model)
        hmm.EM_runOneIterationForList(em, model)
        got, lab = model.estimators.copy(), em.labels()
        orc.set_param_vector(model.param_vector())
        assert orc.run_iteration() == 0
        ref = orc.stats_vector(K)
        assert abs(got[0] - ref[0]) <= LL_RTOL * abs(ref[0]), (got[0], ref[0])
        scale = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
        assert np.all(np.abs(got - ref) <= STAT_RTOL * scale), np.max(np.abs(got - ref) / scale)
        olab = orc.labels()
        assert np.array_equal(lab, olab), f"{np.count_nonzero(lab != olab)} label mismatches of {lab.size}"
    finally:
        em.close()
        orc.close()


def test_full_size_cfg4_one_pass_and_invariants_per_region():
    """BASELINE configs[4] at full size: ONT-R10 preset (hmm_flagger.c:36-58: 8 kb windows, minReadFractionAtEnds
    0.8), 7 bias regions with their own emission series (hmm_utils.c:1605-1652), region changes inside chunks (hmm.c:398-400),
    K = 10: one E-pass with non-trivial parameters against the oracle — labels identical, log-likelihood and every region's
    statistics within 1e-9 — plus the estimator identities per region."""
    store = synth.config(4)
    assert 700_000 < store.n_windows < 850_000 and store.n_regions == 7 and 250 < store.n_chunks < 330
    assert len(np.unique(store.regions())) == 7
    K = hmm.getBestNumberOfCollapsedComps(store)
    assert K == 10
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.ONT_R10_ALPHA)
    em = hmm.EMList(store, model, True, 0.8)
    hmm.EM_runOneIterationForList(em, model)
    hmm.HMM_estimateParameters(model, 1e-3)
    hmm.EM_runOneIterationForList(em, model)
    got = model.estimators.copy()
    lab = em.labels()
    orc = Oracle(store, 0, K, synth.ONT_R10_ALPHA, 0.25, 0.75, True, 0.8, threads=16)
    try:
        orc.set_param_vector(model.param_vector())
        assert orc.run_iteration() == 0
        ref = orc.stats_vector(K)
        assert abs(got[0] - ref[0]) <= LL_RTOL * abs(ref[0]), (got[0], ref[0])
        scale = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
        assert np.all(np.abs(got - ref) <= STAT_RTOL * scale), np.max(np.abs(got - ref) / scale)
        olab = orc.labels()
        assert np.array_equal(lab, olab), f"{np.count_nonzero(lab != olab)} label mismatches of {lab.size}"
        # pairs (i, i+1), i = 1..T-2, are counted in the region of window i+1 (hmm.c:576-590): total xi mass per region = its pairs
        reg = store.regions()
        pairs_r = np.zeros(7)
        for c in range(store.n_chunks):
            a, b = int(store.chunk_off[c]), int(store.chunk_off[c + 1])
            if b - a > 2:
                pairs_r += np.bincount(reg[a + 2:b], minlength=7)
        st = N.region_stride(K)
        for r in range(7):
            base = 1 + r * st
            trans = got[base + 24 * K: base + 24 * K + 16]
            assert pairs_r[r] > 1000 and abs(trans.sum() - pairs_r[r]) <= 1e-7 * pairs_r[r], (r, trans.sum(), pairs_r[r])
            for s_ in (1, 2, 3):
                nc = K if s_ == 3 else 1
                md = got[base + ((s_ * 3 + 0) * 2 + 1) * K: base + ((s_ * 3 + 0) * 2 + 1) * K + nc]
                vd = got[base + ((s_ * 3 + 1) * 2 + 1) * K: base + ((s_ * 3 + 1) * 2 + 1) * K + nc]
                wn = got[base + ((s_ * 3 + 2) * 2 + 0) * K: base + ((s_ * 3 + 2) * 2 + 0) * K + nc]
                wd = got[base + ((s_ * 3 + 2) * 2 + 1) * K: base + ((s_ * 3 + 2) * 2 + 1) * K + nc]
                assert np.array_equal(md, vd) and np.array_equal(md, wn) and np.all(wd == wd[0])
                assert abs(wd[0] - md.sum()) <= 1e-9 * wd[0]
                assert abs(md.sum() - trans.reshape(4, 4)[:, s_].sum()) <= 1e-9 * md.sum()
        post = em.posterior(0, 100_000)
        assert np.allclose(post.sum(axis=1), 1.0, atol=1e-12) and np.array_equal(post.argmax(axis=1), lab[:100_000])
    finally:
        em.close()
        orc.close()


def test_shared_denominator_division_is_bit_identical_where_the_guard_allows_it():
    """k_stats_tile divides many numerators by the same denominator through one refined reciprocal (hf_device.h
    prediv / divp).  Wherever its guard admits the operands the quotient must be the correctly rounded a / d,
    bit for bit; outside (zero / denormal / huge operands) the kernel falls back to the plain division."""
    import ctypes as C
    rng = np.random.default_rng(99)
    n = 2_000_000
    ea, ed = rng.uniform(-400, 400, n), rng.uniform(-400, 400, n)      # the guard admits 2^-380 .. 2^380
    a = rng.uniform(1.0, 2.0, n) * np.exp2(np.floor(ea))
    d = rng.uniform(1.0, 2.0, n) * np.exp2(np.floor(ed))
    a[:1000] = 0.0
    d[1000:2000] = 1e-4                      # terminationProb
    a[2000:3000] = np.nextafter(a[2000:3000], np.inf)
    d[3000:4000] = a[3000:4000]              # quotient exactly 1
    a[4000:4100] = 5e-324                    # denormal: must be rejected by the guard
    d[4100:4200] = 0.0
    fast, exact, safe = np.empty(n), np.empty(n), np.empty(n, dtype=np.int32)
    L = N.lib()
    pd = C.POINTER(C.c_double)
    N.check(L.hf_selftest_division(0, n, a.ctypes.data_as(pd), d.ctypes.data_as(pd), fast.ctypes.data_as(pd),
                                   exact.ctypes.data_as(pd), safe.ctypes.data_as(C.POINTER(C.c_int32))), "hf_selftest_division")
    ok = safe.astype(bool)
    assert ok.sum() > 0.85 * n and not ok[4000:4200].any() and ok[:1000].sum() > 900   # zero numerators are admitted
    bad = ok & (fast.view(np.uint64) != exact.view(np.uint64))
    assert not bad.any(), (int(bad.sum()), np.log2(a[bad][:8]), np.log2(d[bad][:8]), fast[bad][:4], exact[bad][:4])
    assert np.array_equal(exact[ok], a[ok] / d[ok])          # and both are the IEEE quotient


def test_statistics_by_row_agree_with_per_chunk_statistics():
    """HF_STATS_ROWS (hf_rows.h) against HF_STATS_CHUNKS on the same pass: same log-likelihood bits, statistics to
    rounding (a different order of the same additions), same labels / forward / backward; multi-region input with
    contig ends (private rows) and region changes; the mode is reported and can be switched between passes."""
    store = synth.config(4, scale=0.02)
    K = hmm.getBestNumberOfCollapsedComps(store)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.ONT_R10_ALPHA)
    em = hmm.EMList(store, model, True, 0.8)
    try:
        assert em.stats_mode == N.HF_STATS_ROWS
        em.launch(model); rows = em.finish()
        lab_r = em.labels(); f_r, b_r, sc_r = em.forward_backward()
        em.set_stats_mode(N.HF_STATS_CHUNKS)
        em.launch(model); chunks = em.finish()
        lab_c = em.labels(); f_c, b_c, sc_c = em.forward_backward()
        assert abs(rows[0] - chunks[0]) <= 1e-12 * abs(chunks[0])
        scale = np.maximum(np.abs(chunks), 1e-9 * np.abs(chunks).max())
        assert np.all(np.abs(rows - chunks) <= 1e-12 * scale), np.max(np.abs(rows - chunks) / scale)
        assert np.array_equal(rows == 0.0, chunks == 0.0)
        # the two paths cut a chunk differently (segments / tiles): carried-in vectors differ in the last ulp
        assert np.array_equal(lab_r, lab_c) and np.allclose(sc_r, sc_c, rtol=1e-12, atol=0)
        assert np.allclose(f_r, f_c, rtol=1e-11, atol=1e-300) and np.allclose(b_r, b_c, rtol=1e-11, atol=1e-300)
        em.set_stats_mode(N.HF_STATS_ROWS)
        em.launch(model); again = em.finish()
        assert np.array_equal(again, rows)                                 # fixed plan: reproducible bit for bit
    finally:
        em.close()


def test_partial_range_getters_and_rank_total_in_both_statistics_modes():
    """hf_get_forward_backward / hf_get_posterior on a sub-range (pair records in the statistics-by-row mode, lane-minor
    arrays otherwise) equal the slices of the full range; hf_rank_total (what ranks exchange) equals hf_finish's vector;
    hf_last_kernel_ms needs the HF_PROF_PASS bit."""
    store = synth.config(2, scale=0.004)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 4, store, synth.HIFI_ALPHA)
    em = hmm.EMList(store, model)
    other = hmm.EMList(store, model)            # only lends a device buffer: its per-chunk array is the "exchange buffer"
    scratch = other._L.hf_chunk_stats_dev(other._h)
    try:
        for mode in (N.HF_STATS_ROWS, N.HF_STATS_CHUNKS):
            em.set_stats_mode(mode)
            em.launch(model)
            em.rank_total(scratch)
            got = em.finish_gathered(scratch, 0, 1).copy()     # the exchange of a world of one rank
            total = em.finish()
            assert np.array_equal(got, total)
            f, b, sc = em.forward_backward()
            n = store.n_windows
            for first, cnt in ((0, 1), (n - 1, 1), (n // 3, 257), (5, 64)):
                f1, b1, s1 = em.forward_backward(first, cnt)
                assert np.array_equal(f1, f[first:first + cnt]) and np.array_equal(b1, b[first:first + cnt])
                assert np.array_equal(s1, sc[first:first + cnt])
            post = em.posterior()
            assert np.allclose(post.sum(axis=1), 1.0, rtol=0, atol=1e-12)
            assert np.array_equal(em.posterior(7, 30), post[7:37])
        with pytest.raises(N.HFError):
            em.kernel_ms()                     # no pass-level event pair unless profiling asks for it
        em.set_profiling(True)
        em.launch(model); em.finish()
        assert em.kernel_ms() > 0.0
    finally:
        em.close()
        other.close()


def test_sparse_statistics_plan_is_compact():
    """Reads longer than the contigs: every window is a contig-end window with a private emission row and a row of A of its own.
    A plan padded to 64 positions per group would cost 64 positions per pair; hf_create lays the groups out back to back instead
    (rounds 1-2 fell back to the per-chunk statistics here).  Statistics by row == per-chunk statistics == the oracle."""
    store = synth.synthesize([350_000] * 240, 1000, 200_000, [20], seed=5, avg_alignment_len=400_000)   # 84 k windows, none interior
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 4, store, synth.HIFI_ALPHA)
    em = hmm.EMList(store, model)
    try:
        assert em.stats_mode == N.HF_STATS_ROWS
        em.launch(model); rows = np.array(em.finish())
        em.set_stats_mode(N.HF_STATS_CHUNKS)
        em.launch(model); chunks = np.array(em.finish())
        scale = np.maximum(np.abs(chunks), 1e-9 * np.abs(chunks).max())
        assert np.all(np.abs(rows - chunks) <= 1e-12 * scale), np.max(np.abs(rows - chunks) / scale)
        em.set_stats_mode(N.HF_STATS_ROWS)
        em.launch(model); again = np.array(em.finish())
        assert np.array_equal(again, rows)                                 # fixed plan: reproducible bit for bit
    finally:
        em.close()
    for mt, alpha in ((hmm.MODEL_TRUNC_EXP_GAUSSIAN, synth.HIFI_ALPHA), (hmm.MODEL_NEGATIVE_BINOMIAL, np.zeros((4, 4)))):
        _check_pass(store, mt, 4, alpha, N.HF_ALGO_SCAN, n_iter=1)


@pytest.mark.parametrize("plan", ["", "compact", "compact,bpw=3", "padded,bpw=2"])
@pytest.mark.parametrize("seed", range(8))
def test_random_inputs_both_statistics_modes_against_each_other_and_the_oracle(seed, plan, monkeypatch):
    """Seeded random shapes — contig lengths from a few windows to many tiles, window / chunk lengths, 1-5 regions with
    short region runs, clipped windows (End column), read lengths from shorter than a window to longer than a contig,
    K 2..9, the three alpha tables — one full pass: statistics by row == per-chunk statistics to rounding, and the
    per-chunk vector, log-likelihood and labels against the oracle.  `plan`: the layout hf_create chooses by itself, and both
    layouts of the statistics plan forced (HF_STATS_PLAN: compact plans and several batches of row slots per wavefront are
    otherwise chosen for large sparse inputs only)."""
    if plan:
        monkeypatch.setenv("HF_STATS_PLAN", plan)
    rng = np.random.default_rng(1000 + seed)
    window_len = int(rng.choice([500, 1000, 4000]))
    chunk_len = int(rng.choice([20, 77, 300])) * window_len
    n_ctg = int(rng.integers(1, 7))
    lengths = [int(rng.integers(2, 4000)) * window_len + int(rng.integers(0, window_len)) for _ in range(n_ctg)]
    R = int(rng.integers(1, 6))
    region_cov = [int(rng.integers(8, 40)) for _ in range(R)]
    avg_len = int(rng.choice([0, 300, 15_000, 3_000_000]))
    store = synth.synthesize(lengths, window_len, chunk_len, region_cov, seed=seed, avg_alignment_len=avg_len,
                             region_run_bases=(3 * window_len, 200 * window_len))
    clip = np.asarray(store.clip).copy()
    hit = rng.random(clip.size) < 0.03
    clip[hit] = (np.asarray(store.cov)[hit] * 2 + 1).astype(clip.dtype)      # clip ratio >= 1: the End column takes part
    store.clip = clip
    K = int(rng.integers(2, 10))
    alpha = [synth.HIFI_ALPHA, synth.ONT_R10_ALPHA, np.zeros((4, 4))][int(rng.integers(0, 3))]
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, alpha)
    em = hmm.EMList(store, model, bool(rng.integers(0, 2)), 0.9)
    try:
        em.launch(model); a = em.finish(); lab_a = em.labels()
        mode_a = em.stats_mode
        em.set_stats_mode(N.HF_STATS_CHUNKS)
        em.launch(model); b = em.finish(); lab_b = em.labels()
        assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0]) and np.array_equal(lab_a, lab_b)   # per-segment / per-tile sums of the log-likelihood
        scale = np.maximum(np.abs(b), 1e-9 * np.abs(b).max())
        assert np.all(np.abs(a - b) <= 1e-11 * scale), (mode_a, np.max(np.abs(a - b) / scale))
    finally:
        em.close()
    _check_pass(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, alpha, N.HF_ALGO_SCAN, n_iter=1)


_ALTERNATE = """
import sys
import numpy as np
sys.path.insert(0, %r)
from flagger_amd import hmm, synth, _native as N
store = synth.config(2, scale=0.01)
K = 4
model_a = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
model_b = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
em = hmm.EMList(store, model_a)
hmm.EM_runOneIterationForList(em, model_b)
hmm.HMM_estimateParameters(model_b, 1e-3)            # b: one EM step away from a
for mode in (N.HF_STATS_ROWS, N.HF_STATS_CHUNKS):
    em.set_stats_mode(mode)
    em.launch(model_a); ref_a = em.finish().copy()
    em.launch(model_b); ref_b = em.finish().copy()
    assert not np.array_equal(ref_a, ref_b)
    for i in range(300):
        m, ref = (model_a, ref_a) if i %% 2 == 0 else (model_b, ref_b)
        em.launch(m)
        assert np.array_equal(em.finish(), ref), (mode, i)
em.close()
print("alternation ok")
"""


@pytest.mark.parametrize("poll", ["0", "1", "debug"])
def test_completion_never_returns_a_stale_vector(poll):
    """hf_finish synchronises the stream (default) or, with HF_POLL=1, polls a checksummed stamp that the last kernel writes
    after its results: alternate two parameter sets for a few hundred passes — every returned vector must be exactly the one
    of its own parameters, in both statistics modes.  (The switch is read once per process: one subprocess per setting.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _ALTERNATE % root], capture_output=True, text=True, env=dict(os.environ, HF_POLL=poll))
    assert r.returncode == 0 and "alternation ok" in r.stdout, r.stderr[-2000:]
    assert "[poll debug]" not in r.stderr, r.stderr[-2000:]


def _pass_in_subprocess(env, scale, passes=2, multi=False):
    """One or more EM passes of configs[2] x scale in a fresh process (the segment-kernel mode is read at hf_create):
    returns (log-likelihoods, statistics of the last pass, sha1 of the labels, stderr)."""
    import subprocess, sys, json as _json
    code = r"""
import sys, json, hashlib, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from flagger_amd import hmm, synth, _native as N
store = synth.config(2, scale=%r)
K = hmm.getBestNumberOfCollapsedComps(store)
model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
em = hmm.MultiEMList(store, model, 1, exchange=N.HF_EXCHANGE_RANKS) if %r else hmm.EMList(store, model, True, 0.95)
lls = []
for _ in range(%d):
    hmm.EM_runOneIterationForList(em, model)
    lls.append(model.loglikelihood)
    st = np.array(model.estimators, dtype=np.float64).copy()
    hmm.HMM_estimateParameters(model, 1e-3); hmm.HMM_resetEstimators(model)
hmm.EM_runForwardForList(em, model); lls.append(model.loglikelihood)
hmm.EM_runOneIterationForList(em, model)
print(json.dumps({"ll": lls, "stats": st.tolist(), "labels": hashlib.sha1(em.labels().tobytes()).hexdigest()}))
em.close()
""" % (ROOT, os.path.join(ROOT, "tests"), scale, multi, passes)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    return d["ll"], np.array(d["stats"]), d["labels"], r.stderr


@pytest.mark.parametrize("scale", [0.02, 1.0, 3.0], ids=["small", "full", "3x: four rounds of workgroups"])
def test_one_launch_segment_kernel_equals_two_launches(scale):
    """Default: k_seg_fb computes the lane products itself and the segments of a chunk hand their products to each other inside
    the launch (flags, bounded waits: hf_seg.h).  HF_SEG_LAUNCHES=2 is the older k_seg_prod + k_seg_fb pair.  Same arithmetic in
    the same order: log-likelihoods, statistics and labels must be IDENTICAL, also when the grid is several times what the chip
    holds at once (the waits then span dispatch rounds), and nothing may time out."""
    a = _pass_in_subprocess({"HF_SEG_LAUNCHES": "1"}, scale)
    b = _pass_in_subprocess({"HF_SEG_LAUNCHES": "2"}, scale)
    assert "falls back" not in a[3]
    assert a[0] == b[0] and a[2] == b[2] and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("K", [12, 13, 16])
def test_many_collapsed_components_with_and_without_the_kernel_argument_path(K):
    """The words of a one-region parameter block fit k_tables' kernel arguments up to 12 collapsed components (hf_device.h
    HF_KP_MAX_WORDS); 13 and more take the copy ahead of the pass — both against the oracle (the reference's command line clamps K to
    2..10, the library accepts up to HF_MAXCOMP = 16)."""
    store = synth.config(2, scale=0.01)
    _check_pass(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, synth.HIFI_ALPHA, N.HF_ALGO_SCAN, n_iter=2)


@pytest.mark.parametrize("multi", [False, True], ids=["one context", "hf_multi"])
def test_parameter_block_in_the_kernel_arguments_equals_the_copy(multi):
    """One region: the words of the parameter block that are in use travel in k_tables' kernel arguments, every block rebuilds the
    block in LDS and block 0 in global memory for the kernels after it (hf_device.h KParams).  HF_PARAMS_COPY=1 is the copy ahead
    of every pass that other models still take: IDENTICAL log-likelihoods, statistics and labels."""
    a = _pass_in_subprocess({}, 0.05, passes=3, multi=multi)
    b = _pass_in_subprocess({"HF_PARAMS_COPY": "1"}, 0.05, passes=3, multi=multi)
    assert a[0] == b[0] and a[2] == b[2] and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("multi", [False, True], ids=["one context", "hf_multi"])
def test_hand_off_time_out_falls_back_to_two_launches(multi):
    """HF_SEG_TEST_TIMEOUT=1 makes the first one-launch pass wait for flags nobody writes: every wait is given up after its bounded
    number of polls, the flag word carries HF_FLAG_SYNC, the host re-runs the pass with two launches (hf_finish itself; hf_multi
    on HF_E_RETRY) and stays there — the caller sees the results of an ordinary run."""
    a = _pass_in_subprocess({"HF_SEG_TEST_TIMEOUT": "1"}, 0.05, multi=multi)
    b = _pass_in_subprocess({"HF_SEG_LAUNCHES": "2"}, 0.05, multi=multi)
    assert "falls back to k_seg_prod + k_seg_fb" in a[3]
    assert a[0] == b[0] and a[2] == b[2] and np.array_equal(a[1], b[1])


def test_static_guard_of_the_one_launch_hand_off(monkeypatch):
    """a chunk with more segments than the device holds workgroups of k_seg_fb can never have all of them resident —
    every wait of the one-launch kernel would run into its bound.  hf_create sees that and starts such a context in two-launch
    mode (no time-out, no retry): one chunk of 2 M windows (3 907 segments against 12 x 256 resident workgroups), against the oracle;
    and the same decision forced on a small input by pretending a device that holds two workgroups (HF_SEG_RESIDENT)."""
    small = synth.config(2, scale=0.02)
    K = hmm.getBestNumberOfCollapsedComps(small)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, small, synth.HIFI_ALPHA)
    em = hmm.EMList(small, model, True, 0.95)
    assert em.seg_launches == 1
    em.close()
    monkeypatch.setenv("HF_SEG_RESIDENT", "2")              # (this input's longest chunk has three segments)
    em = hmm.EMList(small, model, True, 0.95)
    assert em.seg_launches == 2
    em.close()
    monkeypatch.delenv("HF_SEG_RESIDENT")
    W = 1000
    big = synth.synthesize([2_000_000 * W], W, 2_000_000 * W, [20], seed=21)
    assert big.n_chunks == 1 and big.n_windows == 2_000_000
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 4, big, synth.HIFI_ALPHA)
    em = hmm.EMList(big, model, True, 0.95)
    orc = Oracle(big, hmm.MODEL_TRUNC_EXP_GAUSSIAN, 4, synth.HIFI_ALPHA, 0.25, 0.75, True, 0.95, threads=1)
    try:
        assert em.seg_launches == 2
        hmm.EM_runOneIterationForList(em, model)
        assert em.seg_launches == 2
        assert orc.run_iteration() == 0
        ref, got = orc.stats_vector(model.maxNumberOfComps), model.estimators
        assert abs(got[0] - ref[0]) <= LL_RTOL * abs(ref[0])
        scale = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
        assert np.all(np.abs(got - ref) <= STAT_RTOL * scale)
        assert np.array_equal(em.labels(), orc.labels())
    finally:
        em.close()
        orc.close()


def test_cached_row_blocks_same_bits_and_chosen_by_residency(monkeypatch):
    """VERDICT r04 #4 (the small-shard form of k_seg_fb).  A context with fewer segments than the device holds at twelve workgroups per CU
    keeps the rows of the lane's first nc steps in LDS blocks of their own across the three walks (hf_seg.h) — the same rows, the same
    arithmetic: statistics, labels, forward / backward vectors and a forward-only log-likelihood must be the SAME BITS for every nc, ragged
    segments and several regions included.  hf_create chooses nc from what stays resident: all eight blocks when the device is nearly
    empty, none when the (pretended) device holds just this context's segments at 10.5 KiB each, something in between in between."""
    store = synth.config(4, scale=0.03)                 # 7 regions, ragged chunks, L from 1 to 8
    K = hmm.getBestNumberOfCollapsedComps(store)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.ONT_R10_ALPHA)
    ref = None
    for nc in (0, 1, 3, 5, 8):
        monkeypatch.setenv("HF_SEG_CACHED_STEPS", str(nc))
        em = hmm.EMList(store, model, True, 0.8)
        try:
            assert em.seg_launches == 1 and em.seg_cached_steps == nc
            em.launch(model); st = em.finish().copy()
            lab = em.labels().copy()
            f, b, sc = em.forward_backward()
            em.launch(model, N.HF_MODE_FORWARD_ONLY); fwd = em.finish().copy()
            got = (st, lab, f, b, sc, fwd[0])
            if ref is None:
                ref = got
            else:
                for x, y in zip(ref, got):
                    assert np.array_equal(x, y), nc
        finally:
            em.close()
    monkeypatch.delenv("HF_SEG_CACHED_STEPS")
    em = hmm.EMList(store, model, True, 0.8)
    nseg_small = None
    try:
        assert em.seg_cached_steps == 8                    # a few dozen segments on 256 CUs
    finally:
        em.close()
    # a pretended device (HF_SEG_RESIDENT = workgroups resident at nc = 0; scaled by the real occupancy at every nc)
    chosen = []
    for resident in (1 << 20, 4000, 1200, 600, 300, 100, 12):
        monkeypatch.setenv("HF_SEG_RESIDENT", str(resident))
        em = hmm.EMList(store, model, True, 0.8)
        try:
            chosen.append(em.seg_cached_steps if em.seg_launches == 1 else -1)
        finally:
            em.close()
    monkeypatch.delenv("HF_SEG_RESIDENT")
    assert chosen[0] == 8 and chosen[-1] <= 0, chosen
    assert all(a >= b for a, b in zip(chosen, chosen[1:])), chosen          # less room: no cached blocks (all eight steps or none)
    assert set(chosen) <= {8, 0, -1} and 0 in chosen, chosen
    # two launches (the lane products come from k_seg_prod): nothing is cached
    monkeypatch.setenv("HF_SEG_LAUNCHES", "2")
    em = hmm.EMList(store, model, True, 0.8)
    try:
        assert em.seg_launches == 2 and em.seg_cached_steps == 0
        em.launch(model); st2 = em.finish().copy()
        assert np.array_equal(st2, ref[0]) and np.array_equal(em.labels(), ref[1])
    finally:
        em.close()


def test_host_summed_total_equals_the_device_total_bit_for_bit(monkeypatch):
    """Round 5: on the one-GPU path the blocks of k_row_stats write their partial vectors to pinned host memory and the HOST sums them in the
    order the launch's own last blocks use (hf_estep.hip host_rows_total).  HF_TOTAL=device keeps the in-launch total: statistics, flag
    behaviour and the forward-only log-likelihood must be the same bits — one region and seven, K = 4 / 6 / 10 (three kernel
    instantiations), a sparse plan (four wavefronts per block, several batches per wavefront) — and hf_rank_total after such a pass
    (k_rows_total_late) returns the same vector."""
    cases = [(synth.config(2, scale=0.02), synth.HIFI_ALPHA, 4, 0.95), (synth.config(2, scale=0.05), synth.HIFI_ALPHA, None, 0.95),
             (synth.config(4, scale=0.03), synth.ONT_R10_ALPHA, None, 0.8), (synth.config(5, scale=0.02), synth.ONT_R10_ALPHA, None, 0.8)]
    for store, alpha, K, frac in cases:
        K = hmm.getBestNumberOfCollapsedComps(store) if K is None else K
        model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, alpha)
        got = []
        for mode in ("host", "device"):
            if mode == "device":
                monkeypatch.setenv("HF_TOTAL", "device")
            else:
                monkeypatch.delenv("HF_TOTAL", raising=False)
            em = hmm.EMList(store, model, True, frac)
            other = hmm.EMList(store, model, True, frac)
            scratch = other._L.hf_chunk_stats_dev(other._h)
            try:
                em.launch(model); st = em.finish().copy()
                em.launch(model); em.rank_total(scratch); viax = em.finish_gathered(scratch, 0, 1).copy()
                assert np.array_equal(viax, st), mode
                em.launch(model, N.HF_MODE_FORWARD_ONLY); fwd = em.finish().copy()
                got.append((st, fwd))
            finally:
                em.close(); other.close()
        assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1]), (K, store.n_regions)
    monkeypatch.delenv("HF_TOTAL", raising=False)


def test_sub_passes_through_one_record_buffer(monkeypatch):
    """VERDICT r04 #5 (inputs past the Infinity Cache).  A context with more than ~2.8 M windows runs a full pass in SUB-PASSES of whole chunks —
    k_seg_fb then k_pair_sums per sub-pass through one record buffer that holds a sub-pass at a time (hf_sub_passes).  Forced here on small
    inputs (HF_SUBPASSES): the statistics regroup the pairs per sub-pass, so they equal the one-sub-pass run to rounding (1e-12) and the oracle
    to 1e-9; labels, forward / backward vectors, scales, posteriors and the forward-only log-likelihood are the SAME BITS (the segment kernel does
    not depend on where a record goes); the getters fetch ranges that span sub-passes (the segment kernel runs once more, into the all-windows
    buffer); per-chunk statistics, the negative-binomial model, a compact plan and seven regions take the same route."""
    cases = [(synth.config(2, scale=0.03), hmm.MODEL_TRUNC_EXP_GAUSSIAN, synth.HIFI_ALPHA, None, 0.95, None),
             (synth.config(4, scale=0.03), hmm.MODEL_TRUNC_EXP_GAUSSIAN, synth.ONT_R10_ALPHA, None, 0.8, None),
             (synth.config(5, scale=0.02), hmm.MODEL_TRUNC_EXP_GAUSSIAN, synth.ONT_R10_ALPHA, None, 0.8, "compact"),
             (synth.config(2, scale=0.01), hmm.MODEL_NEGATIVE_BINOMIAL, np.zeros((4, 4)), 5, 0.95, None)]
    for store, mt, alpha, K, frac, plan in cases:
        K = hmm.getBestNumberOfCollapsedComps(store) if K is None else K
        model = hmm.createModel(mt, K, store, alpha)
        if plan:
            monkeypatch.setenv("HF_STATS_PLAN", plan)
        orc = Oracle(store, mt, K, alpha, 0.25, 0.75, True, frac, threads=8)
        try:
            assert orc.run_iteration() == 0
            ref = orc.stats_vector(model.maxNumberOfComps)
            scale = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
            base = None
            for S in (1, 2, 3, 7):
                monkeypatch.setenv("HF_SUBPASSES", str(S))
                em = hmm.EMList(store, model, True, frac)
                try:
                    assert em.sub_passes == min(S, store.n_chunks)
                    assert sum(em.sub_pass_windows(k) for k in range(em.sub_passes)) == store.n_windows
                    em.launch(model); st = em.finish().copy()
                    assert np.all(np.abs(st - ref) <= STAT_RTOL * scale), (S, np.max(np.abs(st - ref) / scale))
                    lab = em.labels().copy()
                    assert np.array_equal(lab, orc.labels()), S
                    n = store.n_windows
                    f, b, sc = em.forward_backward()
                    f1, b1, s1 = em.forward_backward(n // 3, n // 2)                 # a range across sub-pass boundaries
                    assert np.array_equal(f1, f[n // 3:n // 3 + n // 2]) and np.array_equal(b1, b[n // 3:n // 3 + n // 2])
                    post = em.posterior(5, 1000)
                    em.launch(model); st2 = em.finish().copy()                       # a pass after the getters' own run of the segment kernel
                    assert np.array_equal(st2, st)
                    em.set_stats_mode(N.HF_STATS_CHUNKS)
                    em.launch(model); stc = em.finish().copy()
                    assert np.all(np.abs(stc - ref) <= STAT_RTOL * scale), S
                    assert np.array_equal(em.labels(), lab)
                    fc, bc, scc = em.forward_backward(7, 500)
                    assert np.array_equal(fc, f[7:507]) and np.array_equal(bc, b[7:507])
                    em.set_stats_mode(N.HF_STATS_ROWS)
                    em.launch(model, N.HF_MODE_FORWARD_ONLY); fwd = em.finish().copy()
                    got = (lab, f, b, sc, post, fwd[0])
                    if base is None:
                        base, st_base = got, st
                    else:
                        for x, y in zip(base, got):
                            assert np.array_equal(x, y), S
                        assert np.all(np.abs(st - st_base) <= 1e-11 * scale), S
                finally:
                    em.close()
        finally:
            orc.close()
            monkeypatch.delenv("HF_STATS_PLAN", raising=False)
    monkeypatch.delenv("HF_SUBPASSES", raising=False)


def test_device_exp_is_the_hosts_exp_bit_for_bit():
    """csrc/hf_exp.h restates glibc's exp (what the reference's emission densities call: hmm_utils.c:782, 945) with the
    fused multiply-adds of the host's FMA build — the same bits on the device as the same function on the host, and (where the host's libm
    runs its FMA variant: every x86-64 with FMA) as libm's exp itself.  Arguments: what the Gaussian and truncated-exponential densities
    produce (-0.5 d^2 / var, -lambda x: mostly -60..0), the whole finite range, tiny, huge, non-finite.  (Written to test whether the device
    library's exp causes the --accelerate residue: it does not — profiles/r06_exp_fuzz.txt — so the kernels keep the device library's.)"""
    import ctypes as C
    rng = np.random.default_rng(7)
    x = np.concatenate([-rng.uniform(0, 60, 300_000), rng.uniform(-760, 720, 100_000), rng.normal(0, 1e-3, 20_000),
                        -rng.uniform(0, 1, 50_000) ** 4 * 1e-8, rng.integers(0, 2 ** 63, 40_000).astype(np.uint64).view(np.float64),
                        -rng.integers(0, 2 ** 63, 40_000).astype(np.uint64).view(np.float64),
                        np.array([0.0, -0.0, 709.78, 709.79, -708.4, -745.13, -745.14, -1074.0, np.inf, -np.inf, np.nan, 512.0, -512.0, 1024.0, -1024.0])])
    n = x.size
    dev, host, libm = np.empty(n), np.empty(n), np.empty(n)
    pd = C.POINTER(C.c_double)
    N.check(N.lib().hf_selftest_exp(0, n, x.ctypes.data_as(pd), dev.ctypes.data_as(pd), host.ctypes.data_as(pd), libm.ctypes.data_as(pd)), "hf_selftest_exp")
    same = lambda a, b: (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))      # noqa: E731
    bad = ~same(dev, host)
    assert not bad.any(), (int(bad.sum()), x[bad][:5], dev[bad][:5], host[bad][:5])
    if "fma" in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split():
        bad = ~same(dev, libm)
        assert not bad.any(), (int(bad.sum()), x[bad][:5], dev[bad][:5], libm[bad][:5])


def test_check_raises_retry_pass_instead_of_returning_a_stale_result():
    """EMList.check() after a hand-off timed out (HF_SEG_TEST_TIMEOUT: the first one-launch pass waits for flags nobody
    writes) must not look like success — what the caller copied out of that pass is garbage.  It raises RetryPass (the context has
    switched to two launches); the repeated launch + check succeeds and gives the statistics of an ordinary two-launch run."""
    import subprocess, sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from flagger_amd import hmm, synth, _native as N
store = synth.config(2, scale=0.05)
K = hmm.getBestNumberOfCollapsedComps(store)
model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
em = hmm.EMList(store, model, True, 0.95)
em.set_stats_mode(N.HF_STATS_CHUNKS)
assert em.seg_launches == 1
em.launch(model)
try:
    em.check()
    print("NO RETRY")
except hmm.RetryPass as e:
    assert e.code == N.HF_E_RETRY
    assert em.seg_launches == 2
    em.launch(model); em.check()
    a = em.finish().copy()
    em.launch(model); b = em.finish()
    print("RETRY", bool(np.array_equal(a, b)), a[0])
em.close()
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, HF_SEG_TEST_TIMEOUT="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "RETRY True" in r.stdout, (r.stdout, r.stderr[-1000:])


def test_xcd_block_plan_same_bits_and_every_chunk_on_one_residue_class(monkeypatch):
    """VERDICT r05 #2b: HF_SEG_XCD=1 runs the segments through hf_create's block -> segment table: every segment exactly once, all segments
    of a chunk on block indices congruent mod 8, the eight lists within one chunk's segments of each other; same arithmetic, so the same
    bits as block b = segment b; not built where a chunk's segments would span more than half of the resident workgroups (the static guard
    of the one-launch hand-off, extended to the plan); sub-passes have block ranges of their own.  Measured slower at full size
    (profiles/r06_ab_handoff.txt): off unless asked for."""
    store = synth.config(2, scale=0.05)
    K = hmm.getBestNumberOfCollapsedComps(store)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)

    def one(env):
        for k in ("HF_SEG_XCD", "HF_SUBPASSES", "HF_SEG_RESIDENT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        em = hmm.EMList(store, model, True, 0.95)
        try:
            em.launch(model); st = em.finish().copy(); lab = em.labels().copy()
            f, b, sc = em.forward_backward(100, 3000)
            em.launch(model, N.HF_MODE_FORWARD_ONLY); fwd = em.finish()[0]
            return em.seg_xcd_plan, em.seg_block_table(), st, lab, f.copy(), b.copy(), sc.copy(), fwd, em.seg_launches
        finally:
            em.close()
    for env in ({"HF_SEG_XCD": "1"}, {"HF_SEG_XCD": "1", "HF_SUBPASSES": "3"}):
        base = one({k: v for k, v in env.items() if k != "HF_SEG_XCD"})      # (the same sub-passes: their number moves the last bits of a sum)
        assert not base[0] and base[1].size == 0                  # the default: no plan
        got = one(env)
        assert got[0] and got[8] == 1
        tab = got[1]
        segs = tab[tab >= 0]
        assert np.array_equal(np.sort(segs), np.arange(segs.size)) and tab.size % 8 == 0 and tab.size - segs.size < 8 * (3 if "HF_SUBPASSES" in env else 1) * 16
        # segments of a chunk: consecutive indices; chunk boundaries from the store (ceil(T / 512) equal segments per chunk)
        nseg = [int(-(-int(store.chunk_off[c + 1] - store.chunk_off[c]) // 512)) for c in range(store.n_chunks)]
        assert sum(nseg) == segs.size
        block_of = np.empty(segs.size, dtype=np.int64); block_of[tab[tab >= 0]] = np.nonzero(tab >= 0)[0]
        s0 = 0
        for n in nseg:
            res = block_of[s0:s0 + n] % 8
            assert (res == res[0]).all()
            assert (np.diff(block_of[s0:s0 + n]) == 8).all()      # consecutive rows of its list
            s0 += n
        for what, a, b in zip(("statistics", "labels", "f", "b", "scales", "forward-only log-likelihood"), got[2:8], base[2:8]):
            assert np.array_equal(a, b), (env, what)
    # a device that holds fewer than 2 x 8 x (segments of the longest chunk) workgroups: no plan, the one-launch kernel as before
    base = one({})
    got = one({"HF_SEG_XCD": "1", "HF_SEG_RESIDENT": str(8 * 2 * max(nseg) - 1)})
    assert not got[0] and got[8] == 1 and np.array_equal(got[2], base[2])
