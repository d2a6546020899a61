"""N>1 path on CPU: world_size-2 `gloo` run of flagger_amd.dist.ShardedEMList.

The per-rank E-step is replaced by an ORACLE-backed local backend (tests may use the oracle; the product
backend is the HIP EMList), so what is exercised is exactly the distributed logic: chunk sharding, the
all-gather of per-chunk statistic vectors, the ordered global reduction and the replicated M-step.
The sharded result must be bit-identical to a single-process run over all chunks.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from flagger_amd import _native as N
from flagger_amd import dist as fdist
from flagger_amd import hmm, synth
from oracle_py import Oracle

K = 4
ALPHA = synth.HIFI_ALPHA


def _store():
    return synth.synthesize([900_000, 500_000, 120_000, 60_000, 700_000], 1000, 200_000, [20, 26], seed=17,
                            region_run_bases=(20_000, 150_000))


class OracleLocal:
    """Local backend protocol of ShardedEMList backed by one oracle session per chunk."""

    def __init__(self, sub):
        self.sub = sub
        self.orcs = [Oracle(sub.subset_chunks([c]), 0, K, ALPHA, threads=1) for c in range(sub.n_chunks)]

    def launch(self, model, mode):
        v = model.param_vector()
        for o in self.orcs:
            o.set_param_vector(v)
            assert o.run_iteration(forward_only=(mode == N.HF_MODE_FORWARD_ONLY)) == 0

    def chunk_stats_into(self, send):
        for c, o in enumerate(self.orcs):
            send[c] = torch.from_numpy(o.stats_vector(K))

    def reduce_into(self, rows, row_index, n_chunks, total):
        acc = np.zeros(rows.shape[1])
        p = rows.numpy()
        idx = row_index.numpy()
        for c in range(n_chunks):      # list order, as hmm.c:759-763
            acc = acc + p[idx[c]]
        total.copy_(torch.from_numpy(acc))

    def use_rank_totals(self):
        pass

    def rank_total_into(self, send):
        acc = np.zeros(send.shape[0])
        for o in self.orcs:            # this rank's chunks in list order
            acc = acc + o.stats_vector(K)
        send.copy_(torch.from_numpy(acc))

    def check(self):
        pass

    def labels(self):
        return np.concatenate([o.labels() for o in self.orcs]) if self.orcs else np.zeros(0, np.int8)


def _worker(rank, world, port, q, exchange="chunks"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    store = _store()
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, ALPHA)
    V = N.stats_len(model.numberOfRegions, K)
    sh = fdist.ShardedEMList(store, rank, world, OracleLocal, V, exchange=exchange)
    trace = []
    for _ in range(3):
        hmm.EM_runOneIterationForList(sh, model)
        trace.append(model.estimators.copy())
        hmm.HMM_estimateParameters(model, 1e-3)
        hmm.HMM_resetEstimators(model)
    hmm.EM_runForwardForList(sh, model)
    labels = sh.gather_labels()
    q.put((rank, trace, model.param_vector(), labels, model.loglikelihood, sh.bounds))
    tdist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_em_is_bit_identical_to_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference over all chunks
    store = _store()
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, ALPHA)
    orc = Oracle(store, 0, K, ALPHA, threads=2)
    ref_trace = []
    for _ in range(3):
        orc.set_param_vector(model.param_vector())
        assert orc.run_iteration() == 0
        model.estimators = orc.stats_vector(K)
        ref_trace.append(model.estimators.copy())
        hmm.HMM_estimateParameters(model, 1e-3)
        hmm.HMM_resetEstimators(model)
    orc.set_param_vector(model.param_vector())
    assert orc.run_iteration(forward_only=True) == 0
    for rank, trace, pv, labels, ll, bounds in res:
        assert bounds[0] == 0 and bounds[-1] == store.n_chunks and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
        for a, b in zip(trace, ref_trace):
            assert np.array_equal(a, b)                       # statistics bit-identical on every rank
        assert np.array_equal(pv, model.param_vector())       # replicated M-step => identical models
        assert ll == orc.m.contents.loglikelihood
    # labels of the last FULL pass, gathered in global window order
    orc2 = Oracle(store, 0, K, ALPHA, threads=2)
    m2 = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, ALPHA)
    for _ in range(3):
        orc2.set_param_vector(m2.param_vector())
        orc2.run_iteration()
        m2.estimators = orc2.stats_vector(K)
        hmm.HMM_estimateParameters(m2, 1e-3)
    assert np.array_equal(res[0][3], orc2.labels())
    orc.close()
    orc2.close()


def test_shard_bounds_balance_and_edge_cases():
    sizes = np.diff(synth.config(2, scale=0.05).chunk_off)
    for world in (1, 2, 4, 8):
        b = fdist.shard_bounds(sizes, world)
        assert b[0] == 0 and b[-1] == len(sizes) and len(b) == world + 1
        per = [int(sizes[b[r]:b[r + 1]].sum()) for r in range(world)]
        assert sum(per) == int(sizes.sum())
        assert max(per) <= sizes.sum() / world + sizes.max()
    assert fdist.shard_bounds([], 4) == [0, 0, 0, 0, 0]
    assert fdist.shard_bounds([5], 3)[-1] == 1


def test_rank_total_exchange_matches_to_rounding():
    """exchange="ranks": every rank sums its own chunks, one vector per rank is all-gathered and summed in rank order —
    the per-chunk result up to the rounding of a different summation order; every rank holds the same bits."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "ranks")) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    store = _store()
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, K, store, ALPHA)
    orc = Oracle(store, 0, K, ALPHA, threads=2)
    orc.set_param_vector(model.param_vector())
    assert orc.run_iteration() == 0
    ref = orc.stats_vector(K)
    for rank, trace, pv, labels, ll, bounds in res:
        got = trace[0]
        scale = np.maximum(np.abs(ref), 1e-9 * np.abs(ref).max())
        assert np.all(np.abs(got - ref) <= 1e-12 * scale)
        assert np.array_equal(trace[0], res[0][1][0]) and np.array_equal(pv, res[0][2])     # replicas agree bit for bit
    orc.close()


def test_native_shard_bounds_equal_the_python_rule():
    """hf_shard_bounds (what `hmm_flagger --gpus N` shards with) and dist.shard_bounds (what the torch.distributed path
    shards with) cut every chunk list at the same places."""
    import ctypes as C
    L = N.lib()
    rng = np.random.default_rng(1)
    for _ in range(200):
        n = int(rng.integers(0, 40))
        sizes = rng.integers(1, 10000, size=n)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        for world in (1, 2, 3, 5, 8, 13):
            b = (C.c_int32 * (world + 1))()
            assert L.hf_shard_bounds(off.ctypes.data_as(C.POINTER(C.c_int64)), n, world, b) == 0
            assert list(b) == fdist.shard_bounds(sizes, world)
