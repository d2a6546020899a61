"""Multi-GPU E-step: chunks are statically sharded across ranks, one process per GPU, and the
per-chunk sufficient-statistics vectors are exchanged with ONE all-gather per EM iteration over
RCCL/xGMI (torch.distributed backend "nccl"); every rank then sums all chunks in global list order
(hf_reduce_chunks), which makes the statistics — and therefore the whole EM trajectory and the
labels — identical to a 1-GPU run (prefer all-gather + fixed-order summation over
all-reduce at this message size; the collective is latency-bound either way).

The reference has no counterpart: its only parallelism is a pthread pool over chunks followed by
the same in-order reduction (programs/submodules/hmm/hmm.c:739-763).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _native as N
from .synth import WindowStore


def shard_bounds(chunk_sizes: Sequence[int], world: int) -> List[int]:
    """Contiguous runs of the chunk list, balanced by window count: bounds[r]..bounds[r+1] is rank r's run."""
    sizes = np.asarray(chunk_sizes, dtype=np.int64)
    C = len(sizes)
    csum = np.concatenate([[0], np.cumsum(sizes)])
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(csum, target, side="left"))
        # choose the boundary closest to the target, never going backwards
        if k > 0 and abs(csum[k - 1] - target) <= abs(csum[min(k, C)] - target):
            k -= 1
        bounds.append(min(max(k, bounds[-1]), C))
    bounds.append(C)
    return bounds


class ShardedEMList:
    """The EM list of one rank + the exchange.  `make_local(sub_store)` builds the per-rank E-step
    object; the product passes an `hmm.EMList` factory, the CPU/gloo tests pass an oracle-backed one."""

    def __init__(self, store: WindowStore, rank: int, world: int, make_local: Callable, stats_len: int,
                 device=None, group=None, exchange: str = "chunks"):
        import torch
        self.torch = torch
        self.rank, self.world, self.group = rank, world, group
        sizes = np.diff(store.chunk_off)
        self.bounds = shard_bounds(sizes, world)
        self.counts = [self.bounds[r + 1] - self.bounds[r] for r in range(world)]
        self.maxc = max(1, max(self.counts))
        self.n_chunks_total = int(store.n_chunks)
        self.n_windows_total = int(store.n_windows)
        self.local_store = store.subset_chunks(range(self.bounds[rank], self.bounds[rank + 1]))
        self.local = make_local(self.local_store)
        self.V = stats_len
        self.device = device if device is not None else torch.device("cpu")
        # Exchange buffers: rows of V doubles, `rpr` rows per rank, the LAST row of a rank carries its device error-flag
        # word (hf_write_flag_row) so that every rank sees every rank's flags after the one collective of the pass.
        # exchange "chunks": all-gather the per-chunk vectors, sum in global list order (bit-identical to one GPU);
        # "ranks": every rank sums its own chunks (hf_rank_total, statistics by emission row on the HIP backend), one
        # vector per rank is all-gathered and summed in rank order (the same numbers up to the rounding of the order)
        if exchange not in ("chunks", "ranks"):
            raise ValueError("exchange must be 'chunks' or 'ranks'")
        self.exchange = exchange
        self.rpr = (self.maxc if exchange == "chunks" else 1) + 1
        self.flag_row = self.rpr - 1
        self.send = torch.zeros((self.rpr, self.V), dtype=torch.float64, device=self.device)
        self.recv = torch.zeros((world, self.rpr, self.V), dtype=torch.float64, device=self.device)
        if exchange == "chunks":
            # row of global chunk c inside recv viewed as [world*rpr, V]: rank-major, padded to rpr rows per rank
            rows = [r * self.rpr + k for r in range(world) for k in range(self.counts[r])]
            self.n_rows = self.n_chunks_total
            if hasattr(self.local, "bind_chunk_stats"):    # HIP backend: the pass writes its vectors straight into `send`
                self.local.bind_chunk_stats(self.send)
        else:
            rows = [r * self.rpr for r in range(world)]
            self.n_rows = world
            self.local.use_rank_totals()
        self.row_index = torch.tensor(rows if rows else [0], dtype=torch.int32, device=self.device)
        self.force_collective = False      # run the collective even with world == 1 (exercises the RCCL path on one GPU)
        self.total = torch.zeros((self.V,), dtype=torch.float64, device=self.device)

    def run_sharded(self, model, mode: int) -> np.ndarray:
        out = self._run_sharded_once(model, mode)
        if out is None:                   # HF_E_RETRY on every rank (include/hmm_flagger_hip.h): the pass again, in two launches
            out = self._run_sharded_once(model, mode)
        return out

    def _run_sharded_once(self, model, mode: int):
        import torch.distributed as dist
        self.local.launch(model, mode)
        if self.exchange == "ranks":
            self.local.rank_total_into(self.send[0])
        else:
            self.local.chunk_stats_into(self.send)        # no-op when the pass already wrote into `send`
        hip = hasattr(self.local, "finish_exchange")
        if hip:
            self.local.write_flag_row(self.send[self.flag_row])
        if self.world > 1 or self.force_collective:
            dist.all_gather_into_tensor(self.recv.view(-1), self.send.view(-1), group=self.group)
        else:
            self.recv[0].copy_(self.send)
        # every rank sums ALL rows in the fixed order straight out of the gathered buffer
        rows = self.recv.view(self.world * self.rpr, self.V)
        if hip:      # sum into pinned host memory, one sync; the flags of ALL ranks are checked: every rank raises together
            st = self.local.finish_exchange(rows, self.row_index, self.n_rows, self.world, self.rpr, self.flag_row)
            return None if st is None else st.copy()
        self.local.reduce_into(rows, self.row_index, self.n_rows, self.total)
        stats = self.total.cpu().numpy().copy()           # device->host copy synchronises the stream
        try:
            self.local.check()
        except Exception as e:            # hmm.RetryPass: what was gathered and summed above came from a timed-out pass — the whole sequence again
            if getattr(e, "code", None) == N.HF_E_RETRY:
                return None
            raise
        return stats

    def gather_labels(self) -> np.ndarray:
        """Labels of every window in global order (only needed once, for the final BED)."""
        torch = self.torch
        import torch.distributed as dist
        lab = self.local.labels()
        if self.world == 1:
            return lab
        sizes = [0] * self.world
        mine = torch.tensor([lab.size], dtype=torch.int64, device=self.device)
        allsz = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        dist.all_gather(allsz, mine, group=self.group)
        sizes = [int(t.item()) for t in allsz]
        mx = max(1, max(sizes))
        buf = torch.full((mx,), -1, dtype=torch.int8, device=self.device)
        buf[:lab.size] = torch.from_numpy(lab).to(self.device)
        out = [torch.empty(mx, dtype=torch.int8, device=self.device) for _ in range(self.world)]
        dist.all_gather(out, buf, group=self.group)
        return np.concatenate([out[r][:sizes[r]].cpu().numpy() for r in range(self.world)])


class HipLocal:
    """Adapter: hmm.EMList -> the local-backend protocol of ShardedEMList (tensors are torch CUDA tensors
    whose data_ptr() is handed to the C ABI; all work is enqueued on torch's current stream)."""

    def __init__(self, emlist):
        self.em = emlist
        self.em.set_stats_mode(N.HF_STATS_CHUNKS)   # the exchange is made of per-chunk vectors

    def launch(self, model, mode):
        self.em.launch(model, mode)

    def bind_chunk_stats(self, send):
        self.em.bind_chunk_stats(send.data_ptr())

    def chunk_stats_into(self, send):
        self.em.copy_chunk_stats(send.data_ptr())

    def use_rank_totals(self):
        self.em.set_stats_mode(N.HF_STATS_ROWS)     # no per-chunk vectors needed: statistics by emission row

    def rank_total_into(self, send):
        self.em.rank_total(send.data_ptr())

    def write_flag_row(self, row):
        self.em.write_flag_row(row.data_ptr())

    def reduce_into(self, rows, row_index, n_chunks, total):
        self.em.reduce_chunks_indexed(rows.data_ptr(), row_index.data_ptr(), n_chunks, total.data_ptr())

    def finish_exchange(self, rows, row_index, n_rows, world, rows_per_rank, flag_row):
        return self.em.finish_exchange(rows.data_ptr(), row_index.data_ptr(), n_rows, world, rows_per_rank, flag_row)

    def check(self):
        self.em.check()

    def labels(self):
        return self.em.labels()


def make_sharded_hip(store: WindowStore, model, rank: int, world: int, local_rank: int, adjust=True, frac=0.95,
                     algo: int = N.HF_ALGO_SCAN, group=None, exchange: str = "chunks"):
    """One process per GPU: this rank's chunks live on cuda:<local_rank>; kernels and the collective share
    torch's current stream so no extra synchronisation is needed."""
    import torch
    from . import hmm
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def factory(sub):
        return HipLocal(hmm.EMList(sub, model, adjust, frac, device=local_rank, algo=algo, stream=stream))
    V = N.stats_len(model.numberOfRegions, model.maxNumberOfComps)
    return ShardedEMList(store, rank, world, factory, V, device=dev, group=group, exchange=exchange)
