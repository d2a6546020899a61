"""Window table loader / writers of the host library (include/hmm_flagger_io.h): `.cov`, `.cov.gz`, `.bin`
in, `.bin` / final BED / posterior BED out.  Mirrors ChunksCreator_* of the reference
(programs/submodules/chunk/chunk.c) with a run-length-aware parser."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _native as N
from .synth import WindowStore


class Table:
    def __init__(self, path: str, chunk_len: int = 20_000_000, window_len: int = 16000):
        self._L = N.lib()
        self._h = self._L.hfio_load(path.encode(), chunk_len, window_len)
        if not self._h:
            raise OSError(self._L.hfio_last_error().decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None):
            self._L.hfio_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def store(self) -> WindowStore:
        """Copy of the table as the SoA arrays the E-step consumes."""
        L, h = self._L, self._h
        w = N.hf_windows()
        L.hfio_windows(h, C.byref(w))
        n, c = int(w.n_windows), int(w.n_chunks)

        def arr(ptr, count, dtype):
            if count == 0:
                return np.zeros(0, dtype)
            return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)
        nreg = L.hfio_n_regions(h)
        return WindowStore(
            cov=arr(w.cov, n, np.uint16), mapq=arr(w.mapq, n, np.uint16), clip=arr(w.clip, n, np.uint16),
            annot=arr(w.annot, n, np.uint64), truth=arr(L.hfio_truth(h), n, np.int8),
            prediction=arr(L.hfio_prediction(h), n, np.int8), chunk_off=arr(w.chunk_off, c + 1, np.int64),
            chunk_ctg=[L.hfio_chunk_ctg(h, i).decode() for i in range(c)],
            chunk_ctg_len=arr(w.chunk_ctg_len, c, np.int32), chunk_s=arr(w.chunk_s, c, np.int32),
            chunk_e=arr(w.chunk_e, c, np.int32), window_len=L.hfio_window_len(h), chunk_len=L.hfio_chunk_len(h),
            region_coverages=[int(x) for x in arr(L.hfio_region_coverages(h), nreg, np.int32)],
            avg_alignment_len=L.hfio_avg_alignment_len(h),
            annotation_names=tuple(L.hfio_annotation_name(h, i).decode() for i in range(L.hfio_n_annotations(h))),
            start_only=bool(L.hfio_start_only(h)))

    def write_bin(self, path: str) -> None:
        if self._L.hfio_write_bin(self._h, path.encode()) != 0:
            raise OSError(f"cannot write {path}")

    def write_final_bed(self, labels: np.ndarray, path: str, track_name: str = "final_hmm_flagger",
                        min_len_per_state: Sequence[int] = (0, 0, 0, 0)) -> None:
        lab = np.ascontiguousarray(labels, dtype=np.int8)
        ml = (C.c_int32 * 4)(*[int(x) for x in min_len_per_state])
        if self._L.hfio_write_final_bed(self._h, lab.ctypes.data_as(C.POINTER(C.c_int8)), path.encode(),
                                        track_name.encode(), ml) != 0:
            raise OSError(f"cannot write {path}")

    def write_posterior_bed(self, posterior: np.ndarray, labels: np.ndarray, path: str) -> None:
        post = np.ascontiguousarray(posterior, dtype=np.float64)
        lab = np.ascontiguousarray(labels, dtype=np.int8)
        if self._L.hfio_write_posterior_bed(self._h, post.ctypes.data_as(C.POINTER(C.c_double)),
                                            lab.ctypes.data_as(C.POINTER(C.c_int8)), path.encode()) != 0:
            raise OSError(f"cannot write {path}")
