"""Host-side mirror of the reference's HMM/EM interface for the hot path, on top of the C ABI.

Names and call order follow mobinasri/flagger (programs/submodules/hmm/hmm.h:79-113 and
programs/src/hmm_flagger.c:285-488):

    model = createModel(...)                      # hmm_flagger.c:164-237
    emList = EMList(store, model, ...)            # the list of per-chunk EM objects, hmm_flagger.c:320-330
    EM_runOneIterationForList(emList, model)      # hmm.c:739   (HIP kernels)
    converged = HMM_estimateParameters(model, tol)  # hmm.c:120
    HMM_resetEstimators(model)                    # hmm.c:129

One `EMList` is one GPU context holding every chunk of this process; with a process group the
per-chunk statistics of all ranks are all-gathered and summed in global chunk order, so the
result does not depend on the number of GPUs.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N
from .synth import WindowStore

MODEL_TRUNC_EXP_GAUSSIAN = N.HF_MODEL_TRUNC_EXP_GAUSSIAN
MODEL_GAUSSIAN = N.HF_MODEL_GAUSSIAN
MODEL_NEGATIVE_BINOMIAL = N.HF_MODEL_NEGATIVE_BINOMIAL
STATE_NAMES = ("Err", "Dup", "Hap", "Col")


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class HMM:
    """The model (HMM struct, hmm.h:14-25) + the pending sufficient statistics of the last E-step
    (the reference keeps them inside the model's estimator objects)."""

    def __init__(self, handle):
        self._h = handle
        self._L = N.lib()
        self.estimators: Optional[np.ndarray] = None   # reduced statistics vector of the last pass
        self.loglikelihood = 0.0

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.hfm_destroy(self._h)
            self._h = None

    @property
    def numberOfRegions(self) -> int:
        return self._L.hfm_n_regions(self._h)

    @property
    def maxNumberOfComps(self) -> int:
        return self._L.hfm_max_comps(self._h)

    @property
    def modelType(self) -> int:
        return self._L.hfm_model_type(self._h)

    def params(self) -> N.hf_params:
        p = N.hf_params()
        self._L.hfm_params(self._h, C.byref(p))
        return p

    def param_vector(self) -> np.ndarray:
        v = np.empty(self._L.hfm_param_len(self._h), dtype=np.float64)
        self._L.hfm_get_param_vector(self._h, _dptr(v))
        return v

    def set_param_vector(self, v: np.ndarray) -> None:
        v = np.ascontiguousarray(v, dtype=np.float64)
        assert v.size == self._L.hfm_param_len(self._h)
        self._L.hfm_set_param_vector(self._h, _dptr(v))

    def copy(self) -> "HMM":
        m = HMM(self._L.hfm_copy(self._h))
        m.loglikelihood = self.loglikelihood
        return m

    def unpack(self) -> dict:
        """Parameter arrays by name (tests / reporting)."""
        R, K = self.numberOfRegions, N.HF_MAXCOMP
        v = self.param_vector().reshape(R, -1)
        out = {"trans": v[:, :25].reshape(R, 5, 5).copy(), "lambda": v[:, 25].copy(), "trunc_point": v[:, 26].copy()}
        o = 27
        for name in ("mean", "var", "weight"):
            out[name] = v[:, o:o + 4 * K].reshape(R, 4, K).copy()
            o += 4 * K
        return out

    def writeTransitionTsv(self, path: str) -> None:
        if self._L.hfm_write_transition_tsv(self._h, path.encode()) != 0:
            raise OSError(f"cannot write {path}")

    def writeEmissionTsv(self, path: str) -> None:
        if self._L.hfm_write_emission_tsv(self._h, path.encode()) != 0:
            raise OSError(f"cannot write {path}")


def getAlphaMatrix(alphaTsvPath: Optional[str]) -> np.ndarray:
    """hmm_flagger.c:491-515 (None => zeros)."""
    a = np.zeros((4, 4))
    if alphaTsvPath is not None:
        rc = N.lib().hfm_read_alpha_tsv(alphaTsvPath.encode(), _dptr(a))
        if rc == -2:
            raise ValueError(f"There is at least one alpha value in '{alphaTsvPath}' not between 0 and 1.")
        if rc != 0:
            raise OSError(f"cannot read {alphaTsvPath}")
    return a


def getBestNumberOfCollapsedComps(store: WindowStore) -> int:
    """hmm_flagger.c:105-111 with the [2,10] clamp of :1012-1013."""
    cov = np.ascontiguousarray(store.cov, dtype=np.uint16)
    rc = np.asarray(store.region_coverages, dtype=np.int32)
    k = N.lib().hfm_best_collapsed_comps(cov.ctypes.data_as(C.POINTER(C.c_uint16)), cov.size,
                                         rc.ctypes.data_as(C.POINTER(C.c_int32)), rc.size)
    if k < 0:
        raise ValueError("a region coverage of 0 in the header")
    return k


def createModel(modelType: int, numberOfCollapsedComps: int, store: WindowStore, alphaMatrix: np.ndarray,
                maxHighMapqRatio: float = 0.25, minHighMapqRatio: float = 0.75) -> HMM:
    """hmm_flagger.c:164-237 (initialRandomDev = 0)."""
    rc = np.asarray(store.region_coverages, dtype=np.int32)
    a = np.ascontiguousarray(alphaMatrix, dtype=np.float64)
    h = N.lib().hfm_create(modelType, numberOfCollapsedComps, rc.ctypes.data_as(C.POINTER(C.c_int32)), rc.size,
                           int(store.start_only), store.avg_alignment_len, store.window_len, _dptr(a),
                           maxHighMapqRatio, minHighMapqRatio)
    if not h:
        raise ValueError("createModel: bad arguments")
    if modelType == MODEL_NEGATIVE_BINOMIAL and store.n_windows:
        # the per-x tables of the model are rebuilt on the host every iteration: only up to the largest coverage present
        N.lib().hfm_set_max_coverage(h, int(min(int((np.asarray(store.cov) & 0xff).max()), 250)))
    return HMM(h)


def _windows_struct(store: WindowStore, model: "HMM", adjustContigEnds: bool, minReadFractionAtEnds: float):
    """hf_windows over the store's arrays (+ the dict that keeps the contiguous copies alive)."""
    L = N.lib()
    k = dict(
        off=np.ascontiguousarray(store.chunk_off, np.int64), cov=np.ascontiguousarray(store.cov, np.uint16),
        mapq=np.ascontiguousarray(store.mapq, np.uint16), clip=np.ascontiguousarray(store.clip, np.uint16),
        annot=np.ascontiguousarray(store.annot, np.uint64), s=np.ascontiguousarray(store.chunk_s, np.int32),
        e=np.ascontiguousarray(store.chunk_e, np.int32), cl=np.ascontiguousarray(store.chunk_ctg_len, np.int32))
    w = N.hf_windows()
    w.n_windows, w.n_chunks = store.n_windows, store.n_chunks
    w.chunk_off = k["off"].ctypes.data_as(C.POINTER(C.c_int64))
    w.cov = k["cov"].ctypes.data_as(C.POINTER(C.c_uint16))
    w.mapq = k["mapq"].ctypes.data_as(C.POINTER(C.c_uint16))
    w.clip = k["clip"].ctypes.data_as(C.POINTER(C.c_uint16))
    w.annot = k["annot"].ctypes.data_as(C.POINTER(C.c_uint64))
    w.chunk_s = k["s"].ctypes.data_as(C.POINTER(C.c_int32))
    w.chunk_e = k["e"].ctypes.data_as(C.POINTER(C.c_int32))
    w.chunk_ctg_len = k["cl"].ctypes.data_as(C.POINTER(C.c_int32))
    w.window_len, w.mean_read_len = store.window_len, store.avg_alignment_len
    w.adjust_contig_ends, w.min_read_frac = int(adjustContigEnds), float(minReadFractionAtEnds)
    w.max_high_mapq_ratio = L.hfm_max_high_mapq_ratio(model._h)
    w.min_high_mapq_ratio = L.hfm_min_high_mapq_ratio(model._h)
    w.min_highly_clipped_ratio = L.hfm_min_highly_clipped_ratio(model._h)
    return w, k


class RetryPass(N.HFError):
    """HF_E_RETRY out of EMList.check(): repeat the pass and everything that consumed its statistics."""


class MultiHFError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        super().__init__(f"{where} failed with code {code}: {N.lib().hf_multi_last_error().decode(errors='replace')}")


class MultiEMList:
    """The chunk list sharded over the GPUs of this node inside ONE process (include/hmm_flagger_multi.h hf_multi: one
    host thread, stream and RCCL rank per device; one all-gather of statistics per pass).  What `hmm_flagger --gpus N`
    drives; `transport=N.HF_TRANSPORT_LOOPBACK` puts all ranks on one device (tests on a 1-GPU box)."""

    def __init__(self, store: WindowStore, model: "HMM", n_devices: int, adjustContigEnds: bool = True,
                 minReadFractionAtEnds: float = 0.95, devices: Optional[Sequence[int]] = None, algo: int = N.HF_ALGO_SCAN,
                 exchange: int = N.HF_EXCHANGE_CHUNKS, transport: int = N.HF_TRANSPORT_RCCL):
        L = N.lib()
        self._L, self.store = L, store
        w, self._keep = _windows_struct(store, model, adjustContigEnds, minReadFractionAtEnds)
        self.n_regions, self.max_comps = model.numberOfRegions, model.maxNumberOfComps
        self.stats_len = N.stats_len(self.n_regions, self.max_comps)
        dev = (C.c_int * n_devices)(*devices) if devices is not None else None
        h = C.c_void_p()
        rc = L.hf_multi_create(C.byref(w), self.n_regions, self.max_comps, n_devices, dev, algo, exchange, transport, C.byref(h))
        if rc != N.HF_OK:
            raise MultiHFError(rc, "hf_multi_create")
        self._h, self.world = h, n_devices
        self._stats = np.empty(self.stats_len, dtype=np.float64)

    def close(self):
        if getattr(self, "_h", None):
            self._L.hf_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def run_sharded(self, model: "HMM", mode: int) -> np.ndarray:
        p = model.params()
        rc = self._L.hf_multi_estep(self._h, C.byref(p), mode, _dptr(self._stats))
        if rc != N.HF_OK:
            raise MultiHFError(rc, "hf_multi_estep")
        return self._stats.copy()

    def em_iterate(self, model: "HMM", do_mstep: bool = True, tol: float = 1e-3, mode: int = N.HF_MODE_FULL) -> bool:
        """Sharded EM_runOneIterationForList + exchange + HMM_estimateParameters in one native call (hf_multi_em_iterate)."""
        if getattr(self, "_stats_ptr_of", None) is not self._stats:
            self._stats_ptr, self._stats_ptr_of = _dptr(self._stats), self._stats   # (a new ctypes object per .ctypes access: 2 us of host time per call)
            self._cv = C.c_int(0)
            self._cv_ref = C.byref(self._cv)
        rc = self._L.hf_multi_em_iterate(self._h, model._h, mode, int(do_mstep), tol, self._stats_ptr, self._cv_ref)
        if rc != N.HF_OK:
            raise MultiHFError(rc, "hf_multi_em_iterate")
        model.estimators = self._stats
        model.loglikelihood = float(self._stats[0])
        return bool(self._cv.value)

    def rank_stats(self, r: int) -> np.ndarray:
        out = np.empty(self.stats_len, dtype=np.float64)
        N.check(self._L.hf_multi_rank_stats(self._h, r, _dptr(out)), "hf_multi_rank_stats")
        return out

    def shard_sizes(self):
        return [(int(self._L.hf_multi_shard_chunks(self._h, r)), int(self._L.hf_multi_shard_windows(self._h, r))) for r in range(self.world)]

    def labels(self) -> np.ndarray:
        out = np.empty(self.store.n_windows, dtype=np.int8)
        rc = self._L.hf_multi_get_labels(self._h, out.ctypes.data_as(C.POINTER(C.c_int8)))
        if rc != N.HF_OK:
            raise MultiHFError(rc, "hf_multi_get_labels")
        return out

    def posterior(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        n = self.store.n_windows - first if n is None else n
        out = np.empty((n, 4), dtype=np.float64)
        rc = self._L.hf_multi_get_posterior(self._h, first, n, _dptr(out))
        if rc != N.HF_OK:
            raise MultiHFError(rc, "hf_multi_get_posterior")
        return out


HF_COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """Rank 0's RCCL id for `RankEMList` (include/hmm_flagger_multi.h hf_comm_unique_id); the launcher carries it to the others."""
    buf = C.create_string_buffer(HF_COMM_ID_BYTES)
    N.check(N.lib().hf_comm_unique_id(buf), "hf_comm_unique_id")
    return buf.raw


class RankEMList(MultiEMList):
    """One PROCESS per GPU (torch.distributed.run): this process is rank `rank` of `world` on `device` and holds its shard of
    the chunk list; pass + exchange (RCCL all-gather on the pass's own stream) + ordered reduction are ONE native call per
    EM pass (hf_multi_create_rank / hf_multi_estep).  Collective: every rank constructs it with the same store."""

    def __init__(self, store: WindowStore, model: "HMM", world: int, rank: int, device: int, unique_id: bytes,
                 adjustContigEnds: bool = True, minReadFractionAtEnds: float = 0.95, algo: int = N.HF_ALGO_SCAN,
                 exchange: int = N.HF_EXCHANGE_CHUNKS):
        L = N.lib()
        self._L, self.store = L, store
        w, self._keep = _windows_struct(store, model, adjustContigEnds, minReadFractionAtEnds)
        self.n_regions, self.max_comps = model.numberOfRegions, model.maxNumberOfComps
        self.stats_len = N.stats_len(self.n_regions, self.max_comps)
        if len(unique_id) != HF_COMM_ID_BYTES:
            raise ValueError("unique_id must be the %d bytes of comm_unique_id()" % HF_COMM_ID_BYTES)
        self._id = C.create_string_buffer(unique_id, HF_COMM_ID_BYTES)
        h = C.c_void_p()
        rc = L.hf_multi_create_rank(C.byref(w), self.n_regions, self.max_comps, world, rank, device, algo, exchange, self._id, C.byref(h))
        if rc != N.HF_OK:
            raise MultiHFError(rc, "hf_multi_create_rank")
        self._h, self.world, self.rank = h, world, rank
        self._stats = np.empty(self.stats_len, dtype=np.float64)
        self.first_window = int(L.hf_multi_local_first_window(h))
        self.n_local_windows = int(L.hf_multi_local_windows(h))
        # borrowed view of the rank's context: profiling switches and kernel times (bench.py)
        self.em = EMList.__new__(EMList)
        self.em._L, self.em.store, self.em._h, self.em._borrowed = L, None, C.c_void_p(L.hf_multi_local_ctx(h)), True
        self.em.n_regions, self.em.max_comps, self.em.stats_len = self.n_regions, self.max_comps, self.stats_len

    def close(self):
        if getattr(self, "em", None) is not None:
            self.em._h = None
        super().close()

    def labels(self) -> np.ndarray:
        """Labels of ALL windows' positions, but only this rank's range is known here: the others are -1 (hf_multi_get_labels
        fills this rank's windows at their global positions).  Use local_labels() for the rank's own slice."""
        out = np.full(self.store.n_windows, -1, dtype=np.int8)
        rc = self._L.hf_multi_get_labels(self._h, out.ctypes.data_as(C.POINTER(C.c_int8)))
        if rc != N.HF_OK:
            raise MultiHFError(rc, "hf_multi_get_labels")
        return out

    def posterior(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        """Posteriors of windows first .. first + n; rows of windows that other ranks hold are NaN."""
        n = self.store.n_windows - first if n is None else n
        out = np.full((n, 4), np.nan, dtype=np.float64)
        rc = self._L.hf_multi_get_posterior(self._h, first, n, _dptr(out))
        if rc != N.HF_OK:
            raise MultiHFError(rc, "hf_multi_get_posterior")
        return out

    def local_labels(self) -> np.ndarray:
        """Labels of this rank's windows (global positions first_window .. first_window + n_local_windows)."""
        return self.labels()[self.first_window:self.first_window + self.n_local_windows]


class EMList:
    """All per-chunk EM objects of this process (stList<EM*> in the reference) as ONE device context:
    the windows are uploaded once and stay resident in HBM (hf_create)."""

    def __init__(self, store: WindowStore, model: HMM, adjustContigEnds: bool = True,
                 minReadFractionAtEnds: float = 0.95, device: int = 0, algo: int = N.HF_ALGO_SCAN,
                 stream: int = 0):
        L = N.lib()
        self._L = L
        self.store = store
        self.stream = C.c_void_p(stream)
        w, self._keep = _windows_struct(store, model, adjustContigEnds, minReadFractionAtEnds)
        self.n_regions, self.max_comps = model.numberOfRegions, model.maxNumberOfComps
        self.stats_len = N.stats_len(self.n_regions, self.max_comps)
        h = C.c_void_p()
        N.check(L.hf_create(C.byref(w), self.n_regions, self.max_comps, device, algo, C.byref(h)), "hf_create")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_borrowed", False):      # a view of a context something else owns (RankEMList.em)
            self._h = None
            return
        if getattr(self, "_h", None):
            self._L.hf_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_stats_mode(self, mode: int) -> None:
        """N.HF_STATS_ROWS (default where it applies) or N.HF_STATS_CHUNKS (per-chunk vectors, needed by the multi-GPU exchange)."""
        N.check(self._L.hf_set_stats_mode(self._h, int(mode)), "hf_set_stats_mode")

    @property
    def stats_mode(self) -> int:
        return int(self._L.hf_get_stats_mode(self._h))

    @property
    def seg_launches(self) -> int:
        """1: one-launch segment kernel, 2: k_seg_prod + k_seg_fb, 0: no segment kernels (hf_seg_launches)."""
        return int(self._L.hf_seg_launches(self._h))

    @property
    def seg_cached_steps(self) -> int:
        """Row blocks a segment workgroup keeps in LDS across its three walks (hf_seg_cached_steps; 0 on a device full of segments)."""
        return int(self._L.hf_seg_cached_steps(self._h))

    @property
    def sub_passes(self) -> int:
        """Sub-passes of a full pass (hf_sub_passes: 1 unless the pair records would not fit the Infinity Cache)."""
        return int(self._L.hf_sub_passes(self._h)) if hasattr(self._L, "hf_sub_passes") else 1

    def sub_pass_windows(self, k: int = 0) -> int:
        return int(self._L.hf_sub_pass_windows(self._h, int(k)))

    @property
    def seg_xcd_plan(self) -> bool:
        """True when the one-launch segment kernel runs its blocks through hf_create's block -> segment table (hf_seg_xcd_plan)."""
        return bool(self._L.hf_seg_xcd_plan(self._h)) if hasattr(self._L, "hf_seg_xcd_plan") else False

    def seg_block_table(self) -> np.ndarray:
        n = int(self._L.hf_seg_block_table(self._h, None, 0))
        out = np.empty(n, dtype=np.int32)
        self._L.hf_seg_block_table(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out

    def create_phases(self) -> dict:
        """{phase: ms} of this context's hf_create, in call order; the last entry is "total" (hf_create_phases)."""
        if not hasattr(self._L, "hf_create_phases"):
            return {}
        ms, names = (C.c_double * 64)(), (C.c_char_p * 64)()
        n = min(int(self._L.hf_create_phases(self._h, 64, ms, names)), 64)
        out = {}
        for i in range(n):
            k = names[i].decode()
            out[k] = out.get(k, 0.0) + float(ms[i])
        return out

    # --- E-step pieces (used directly by the multi-GPU path in dist.py) ---
    def launch(self, model: HMM, mode: int = N.HF_MODE_FULL) -> None:
        p = model.params()
        N.check(self._L.hf_estep(self._h, C.byref(p), mode, self.stream), "hf_estep")

    def finish(self) -> np.ndarray:
        out = np.empty(self.stats_len, dtype=np.float64)
        N.check(self._L.hf_finish(self._h, _dptr(out), self.stream), "hf_finish")
        return out

    def check(self) -> None:
        """hf_check: the error flags of the last pass.  HF_E_RETRY (a hand-off of the one-launch segment kernel timed out; the context has
        switched to two launches) is not an error of the data, but what the timed-out pass left behind — statistics the caller has already
        copied, reduced or all-gathered — is garbage: it is raised as `RetryPass` so that the caller repeats its WHOLE sequence
        (launch, exchange, reduction, check), as ShardedEMList.run_sharded does for all ranks together; hf_finish repeats the pass inside
        the library, where it reads the result itself.  (round 5 re-launched here and returned success over the stale copy.)"""
        rc = self._L.hf_check(self._h, self.stream)
        if rc == N.HF_E_RETRY:
            raise RetryPass(rc, "hf_check")
        N.check(rc, "hf_check")

    def rank_total(self, dst_dev_ptr: int) -> None:
        """This context's statistics vector (either statistics mode) into device memory."""
        N.check(self._L.hf_rank_total(self._h, C.c_void_p(dst_dev_ptr), self.stream), "hf_rank_total")

    def copy_chunk_stats(self, dst_dev_ptr: int) -> None:
        N.check(self._L.hf_copy_chunk_stats(self._h, C.c_void_p(dst_dev_ptr), self.stream), "hf_copy_chunk_stats")

    def finish_gathered(self, rows_dev_ptr: int, row_index_dev_ptr: int, n_chunks: int) -> np.ndarray:
        """Fixed-order sum of the all-gathered per-chunk vectors into host memory + flag check: one sync (hf_finish_gathered)."""
        if not hasattr(self, "_stats_buf"):
            self._stats_buf = np.empty(self.stats_len, dtype=np.float64)
        N.check(self._L.hf_finish_gathered(self._h, C.c_void_p(rows_dev_ptr), C.c_void_p(row_index_dev_ptr) if row_index_dev_ptr else None, n_chunks,
                                           _dptr(self._stats_buf), self.stream), "hf_finish_gathered")
        return self._stats_buf

    def finish_exchange(self, rows_dev_ptr: int, row_index_dev_ptr: int, n_rows: int, world: int, rows_per_rank: int,
                        flag_row: int) -> np.ndarray:
        """hf_finish_exchange: hf_finish_gathered + the error flags of every rank (row `flag_row` of each rank's rows)."""
        if not hasattr(self, "_stats_buf"):
            self._stats_buf = np.empty(self.stats_len, dtype=np.float64)
        rc = self._L.hf_finish_exchange(self._h, C.c_void_p(rows_dev_ptr), C.c_void_p(row_index_dev_ptr), n_rows, world,
                                        rows_per_rank, flag_row, _dptr(self._stats_buf), self.stream)
        if rc == N.HF_E_RETRY:          # every rank gets it together: the caller runs the pass again (ShardedEMList.run_sharded)
            return None
        N.check(rc, "hf_finish_exchange")
        return self._stats_buf

    def bind_chunk_stats(self, rows_dev_ptr: int) -> None:
        N.check(self._L.hf_bind_chunk_stats(self._h, C.c_void_p(rows_dev_ptr)), "hf_bind_chunk_stats")

    def write_flag_row(self, row_dev_ptr: int) -> None:
        N.check(self._L.hf_write_flag_row(self._h, C.c_void_p(row_dev_ptr), self.stream), "hf_write_flag_row")

    def reduce_chunks_indexed(self, rows_dev_ptr: int, row_index_dev_ptr: int, n_chunks: int, dst_dev_ptr: int) -> None:
        N.check(self._L.hf_reduce_chunks_indexed(self._h, C.c_void_p(rows_dev_ptr), C.c_void_p(row_index_dev_ptr) if row_index_dev_ptr else None, n_chunks,
                                                 C.c_void_p(dst_dev_ptr), self.stream), "hf_reduce_chunks_indexed")

    def reduce_chunks(self, src_dev_ptr: int, n_chunks: int, dst_dev_ptr: int) -> None:
        N.check(self._L.hf_reduce_chunks(self._h, C.c_void_p(src_dev_ptr), n_chunks, C.c_void_p(dst_dev_ptr),
                                         self.stream), "hf_reduce_chunks")

    def em_iterate(self, model: "HMM", do_mstep: bool = True, tol: float = 1e-3, mode: int = N.HF_MODE_FULL) -> bool:
        """EM_runOneIterationForList + HMM_estimateParameters in one native call (hf_em_iterate).  Leaves the
        statistics in model.estimators; returns the convergence flag of the M-step."""
        if not hasattr(self, "_stats_buf"):
            self._stats_buf = np.empty(self.stats_len, dtype=np.float64)
            self._stats_ptr = _dptr(self._stats_buf)       # (numpy builds a new ctypes object per .ctypes access: 2 us of host time per call)
            self._cv = C.c_int(0)
            self._cv_ref = C.byref(self._cv)
        rc = self._L.hf_em_iterate(self._h, model._h, mode, int(do_mstep), tol, self._stats_ptr, self._cv_ref, self.stream)
        if rc != N.HF_OK:
            N.check(rc, "hf_em_iterate")
        model.estimators = self._stats_buf
        model.loglikelihood = float(self._stats_buf[0])
        return bool(self._cv.value)

    def kernel_ms(self) -> float:
        ms = C.c_float()
        N.check(self._L.hf_last_kernel_ms(self._h, C.byref(ms)), "hf_last_kernel_ms")
        return float(ms.value)

    def set_profiling(self, kernels=True) -> None:
        """True/False: all kernels / none; or an iterable of kernel names (hf_kernel_name) to bracket with events."""
        names = [self._L.hf_kernel_name(i).decode() for i in range(N.HF_NKERNELS)]
        if kernels is True:
            mask = ((1 << N.HF_NKERNELS) - 1) | N.HF_PROF_PASS      # every kernel + the whole pass (kernel_ms)
        elif not kernels:
            mask = 0
        else:
            mask = 0
            for k in kernels:
                mask |= 1 << names.index(k)
        N.check(self._L.hf_set_profiling(self._h, mask), "hf_set_profiling")

    def set_profiling_stride(self, every_nth_pass: int) -> None:
        """Bracket the selected kernels in every n-th pass only (hf_set_profiling_stride)."""
        N.check(self._L.hf_set_profiling_stride(self._h, int(every_nth_pass)), "hf_set_profiling_stride")

    def kernel_times(self) -> dict:
        """Duration (ms) of each selected kernel in the last pass, from HIP events on the launch stream."""
        ms = (C.c_float * N.HF_NKERNELS)()
        N.check(self._L.hf_kernel_times(self._h, ms), "hf_kernel_times")
        return {self._L.hf_kernel_name(i).decode(): float(ms[i]) for i in range(N.HF_NKERNELS)}

    def kernel_time_sums(self) -> dict:
        """{kernel: (total ms, launches)} accumulated by the library since set_profiling (HIP events on the launch stream)."""
        sums = (C.c_double * N.HF_NKERNELS)()
        cnt = (C.c_int64 * N.HF_NKERNELS)()
        N.check(self._L.hf_kernel_time_sums(self._h, sums, cnt), "hf_kernel_time_sums")
        return {self._L.hf_kernel_name(i).decode(): (float(sums[i]), int(cnt[i])) for i in range(N.HF_NKERNELS)}

    # --- results ---
    def labels(self) -> np.ndarray:
        out = np.empty(self.store.n_windows, dtype=np.int8)
        N.check(self._L.hf_get_labels(self._h, out.ctypes.data_as(C.POINTER(C.c_int8))), "hf_get_labels")
        return out

    def posterior(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        """EM_getPosterior for windows [first, first+n) — hmm.c:671-685."""
        n = self.store.n_windows - first if n is None else n
        out = np.empty((n, 4), dtype=np.float64)
        N.check(self._L.hf_get_posterior(self._h, first, n, _dptr(out)), "hf_get_posterior")
        return out

    def forward_backward(self, first: int = 0, n: Optional[int] = None):
        n = self.store.n_windows - first if n is None else n
        f, b, sc = np.empty((n, 4)), np.empty((n, 4)), np.empty(n)
        N.check(self._L.hf_get_forward_backward(self._h, first, n, _dptr(f), _dptr(b), _dptr(sc)),
                "hf_get_forward_backward")
        return f, b, sc


def EM_runOneIterationForList(emList, model: HMM, threads: int = 0) -> None:
    """hmm.c:739-763.  `threads` is accepted for signature compatibility; the result never depended
    on it.  `emList` is an EMList or a dist.ShardedEMList."""
    if hasattr(emList, "run_sharded"):
        stats = emList.run_sharded(model, N.HF_MODE_FULL)
    else:
        emList.launch(model, N.HF_MODE_FULL)
        stats = emList.finish()
    model.estimators = stats
    model.loglikelihood = float(stats[0])


def EM_runForwardForList(emList, model: HMM, threads: int = 0) -> None:
    """hmm.c:790-816 — forward only; sets model.loglikelihood."""
    if hasattr(emList, "run_sharded"):
        stats = emList.run_sharded(model, N.HF_MODE_FORWARD_ONLY)
    else:
        emList.launch(model, N.HF_MODE_FORWARD_ONLY)
        stats = emList.finish()
    model.loglikelihood = float(stats[0])


def HMM_estimateParameters(model: HMM, convergenceTol: float) -> bool:
    """hmm.c:120-127; consumes the statistics left by EM_runOneIterationForList."""
    if model.estimators is None:
        raise RuntimeError("HMM_estimateParameters: no statistics (run EM_runOneIterationForList first)")
    st = np.ascontiguousarray(model.estimators, dtype=np.float64)
    return bool(N.lib().hfm_estimate(model._h, _dptr(st), float(convergenceTol)))


def HMM_resetEstimators(model: HMM) -> None:
    """hmm.c:129-134."""
    model.estimators = None


def runHMMFlagger(emList, model: HMM, numberOfIterations: int = 100, convergenceTol: float = 0.001,
                  outputDir: Optional[str] = None, writeParameterStatsPerIteration: bool = False,
                  is_writer: bool = True) -> List[float]:
    """EM outer loop of hmm_flagger.c:285-488 (no --accelerate here: that loop is flagger_amd/csrc/hf_squarem.h, driven by the command line).  Returns the
    log-likelihood of every E-pass (the rows of loglikelihood.tsv)."""
    lls: List[float] = []
    llf = None
    write = outputDir is not None and is_writer
    if write:
        llf = open(os.path.join(outputDir, "loglikelihood.tsv"), "w")
        llf.write("#Iteration\tEffective_Iteration\tLoglikelihood\n")
        model.writeTransitionTsv(os.path.join(outputDir, "transition_initial.tsv"))
        model.writeEmissionTsv(os.path.join(outputDir, "emission_initial.tsv"))
    it, converged = 1, False
    fused = hasattr(emList, "em_iterate")      # single GPU: E-step + M-step in one native call
    while it <= numberOfIterations and not converged:
        if fused:
            converged = emList.em_iterate(model, True, convergenceTol)
        else:
            EM_runOneIterationForList(emList, model)
        lls.append(model.loglikelihood)
        if llf:
            llf.write("%d\t%d\t%.4f\n" % (it - 1, it - 1, model.loglikelihood))
        if not fused:
            converged = HMM_estimateParameters(model, convergenceTol)
        HMM_resetEstimators(model)
        if write and writeParameterStatsPerIteration:
            model.writeTransitionTsv(os.path.join(outputDir, f"transition_iteration_{it}.tsv"))
            model.writeEmissionTsv(os.path.join(outputDir, f"emission_iteration_{it}.tsv"))
        it += 1
    EM_runOneIterationForList(emList, model)   # final inference, hmm_flagger.c:464
    lls.append(model.loglikelihood)
    if llf:
        llf.write("%d\t%d\t%.4f\n" % (it - 1, it - 1, model.loglikelihood))
        llf.close()
        model.writeTransitionTsv(os.path.join(outputDir, "transition_final.tsv"))
        model.writeEmissionTsv(os.path.join(outputDir, "emission_final.tsv"))
    return lls
