"""ctypes driver of oracle/liboracle_hf.so — TEST INFRASTRUCTURE ONLY (the checker, never the product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
MAXC = 16


class OEst(C.Structure):
    _fields_ = [("num", (C.c_double * MAXC) * 3), ("den", (C.c_double * MAXC) * 3)]


class ORegion(C.Structure):
    _fields_ = [("trans", (C.c_double * 5) * 5), ("pseudo", (C.c_double * 5) * 5), ("count", (C.c_double * 5) * 5),
                ("lambda_", C.c_double), ("trunc_point", C.c_double),
                ("mean", (C.c_double * MAXC) * 4), ("var", (C.c_double * MAXC) * 4), ("weight", (C.c_double * MAXC) * 4),
                ("theta", (C.c_double * MAXC) * 4), ("nb_lambda", (C.c_double * MAXC) * 4),
                ("est", OEst * 4)]


class OModel(C.Structure):
    _fields_ = [("model_type", C.c_int), ("n_regions", C.c_int), ("ncomp", C.c_int * 4),
                ("alpha", (C.c_double * 4) * 4), ("max_high_mapq_ratio", C.c_double),
                ("min_high_mapq_ratio", C.c_double), ("min_highly_clipped_ratio", C.c_double),
                ("loglikelihood", C.c_double), ("regions", C.POINTER(ORegion))]


class OChunk(C.Structure):
    _fields_ = [("ctg", C.c_char * 200), ("ctg_len", C.c_int32), ("s", C.c_int32), ("e", C.c_int32), ("n", C.c_int32),
                ("cov", C.POINTER(C.c_uint16)), ("mapq", C.POINTER(C.c_uint16)), ("clip", C.POINTER(C.c_uint16)),
                ("annot", C.POINTER(C.c_uint64)), ("truth", C.POINTER(C.c_int8)), ("prediction", C.POINTER(C.c_int8)),
                ("f", C.POINTER(C.c_double)), ("b", C.POINTER(C.c_double)), ("scales", C.POINTER(C.c_double)),
                ("loglikelihood", C.c_double)]


class OChunks(C.Structure):
    _fields_ = [("n_annotations", C.c_int32), ("annotation_names", C.POINTER(C.c_char_p)), ("n_regions", C.c_int32),
                ("region_coverages", C.c_int32 * 64), ("n_labels", C.c_int32), ("truth_available", C.c_bool),
                ("prediction_available", C.c_bool), ("start_only", C.c_bool), ("avg_alignment_len", C.c_int32),
                ("chunk_len", C.c_int32), ("window_len", C.c_int32), ("n_chunks", C.c_int32),
                ("chunks", C.POINTER(OChunk))]


class ORunOpts(C.Structure):
    _fields_ = [("adjust_contig_ends", C.c_bool), ("min_read_frac", C.c_double), ("mean_read_len", C.c_int),
                ("threads", C.c_int)]


class OEmOpts(C.Structure):
    _fields_ = [("iterations", C.c_int), ("tol", C.c_double), ("write_params_per_iter", C.c_bool),
                ("write_posterior", C.c_bool), ("out_dir", C.c_char_p), ("accelerate", C.c_bool)]


_lib = None


def build() -> None:
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle_hf.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.ohf_read_bin.restype = C.POINTER(OChunks)
        L.ohf_read_bin.argtypes = [C.c_char_p]
        L.ohf_chunks_destroy.argtypes = [C.POINTER(OChunks)]
        L.ohf_model_create.restype = C.POINTER(OModel)
        L.ohf_model_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_bool, C.c_int, C.c_int,
                                       C.POINTER(C.c_double), C.c_double, C.c_double]
        L.ohf_model_destroy.argtypes = [C.POINTER(OModel)]
        L.ohf_run_iteration.restype = C.c_int
        L.ohf_run_iteration.argtypes = [C.POINTER(OChunks), C.POINTER(OModel), C.POINTER(ORunOpts), C.c_int]
        L.ohf_estimate_parameters.restype = C.c_bool
        L.ohf_estimate_parameters.argtypes = [C.POINTER(OModel), C.c_double]
        L.ohf_reset_estimators.argtypes = [C.POINTER(OModel)]
        L.ohf_best_collapsed_comps.restype = C.c_int
        L.ohf_best_collapsed_comps.argtypes = [C.POINTER(OChunks)]
        L.ohf_run_em.restype = C.c_int
        L.ohf_run_em.argtypes = [C.POINTER(OChunks), C.POINTER(OModel), C.POINTER(ORunOpts), C.POINTER(OEmOpts),
                                 C.POINTER(C.c_double), C.c_int]
        L.ohf_beta.restype = C.c_double
        L.ohf_beta.argtypes = [C.POINTER(OChunk), C.c_int, C.c_int, C.POINTER(ORunOpts)]
        L.ohf_write_final_bed.restype = C.c_int
        L.ohf_write_final_bed.argtypes = [C.POINTER(OChunks), C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        L.ohf_estimate_lambda.restype = C.c_double
        L.ohf_estimate_lambda.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double]
        _lib = L
    return _lib


class Oracle:
    """One oracle session over a WindowStore (round-tripped through the reference's .bin format)."""

    def __init__(self, store, model_type=0, n_collapsed=None, alpha=None, max_mapq=0.25, min_mapq=0.75,
                 adjust=True, min_read_frac=0.95, threads=4):
        L = lib()
        self.L = L
        self.store = store
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as tf:
            path = tf.name
        try:
            store.write_bin(path)
            self.cc = L.ohf_read_bin(path.encode())
        finally:
            os.unlink(path)
        assert self.cc, "oracle could not read the .bin"
        if n_collapsed is None:
            n_collapsed = L.ohf_best_collapsed_comps(self.cc)
        self.K = n_collapsed
        a = np.zeros((4, 4)) if alpha is None else np.ascontiguousarray(alpha, dtype=np.float64)
        rc = np.asarray(store.region_coverages, dtype=np.int32)
        self.m = L.ohf_model_create(model_type, n_collapsed, rc.ctypes.data_as(C.POINTER(C.c_int32)), rc.size,
                                    bool(store.start_only), store.avg_alignment_len, store.window_len,
                                    a.ctypes.data_as(C.POINTER(C.c_double)), max_mapq, min_mapq)
        assert self.m, "oracle could not create the model"
        self.opts = ORunOpts(adjust, min_read_frac, store.avg_alignment_len, threads)
        self.model_type = model_type

    def close(self):
        if self.cc:
            self.L.ohf_chunks_destroy(self.cc)
            self.cc = None
        if self.m:
            self.L.ohf_model_destroy(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _param_fields(self):
        # negative_binomial: the product keeps theta / lambda in the mean / var slots of its vector
        return ("theta", "nb_lambda", "weight") if self.model_type == 2 else ("mean", "var", "weight")

    # ---- parameters <-> the product's flat vector layout (hfm_get_param_vector) ----
    def param_vector(self) -> np.ndarray:
        R = self.m.contents.n_regions
        out = []
        for r in range(R):
            g = self.m.contents.regions[r]
            out.append(np.ctypeslib.as_array(g.trans).ravel().copy())
            out.append(np.array([g.lambda_, g.trunc_point]))
            for name in self._param_fields():
                out.append(np.ctypeslib.as_array(getattr(g, name)).ravel().copy())
        return np.concatenate(out)

    def set_param_vector(self, v: np.ndarray) -> None:
        R = self.m.contents.n_regions
        v = np.asarray(v, dtype=np.float64).reshape(R, -1)
        for r in range(R):
            g = self.m.contents.regions[r]
            np.ctypeslib.as_array(g.trans)[:] = v[r, :25].reshape(5, 5)
            g.lambda_, g.trunc_point = float(v[r, 25]), float(v[r, 26])
            o = 27
            for name in self._param_fields():
                np.ctypeslib.as_array(getattr(g, name))[:] = v[r, o:o + 4 * MAXC].reshape(4, MAXC)
                o += 4 * MAXC

    # ---- one E-pass ----
    def run_iteration(self, forward_only=False) -> int:
        self.L.ohf_reset_estimators(self.m)
        return self.L.ohf_run_iteration(self.cc, self.m, C.byref(self.opts), int(forward_only))

    def stats_vector(self, K: int) -> np.ndarray:
        """Model estimators in the layout of include/hmm_flagger_hip.h."""
        R = self.m.contents.n_regions
        stride = 24 * K + 16
        out = np.zeros(1 + R * stride)
        out[0] = self.m.contents.loglikelihood
        ncomp = list(self.m.contents.ncomp)
        for r in range(R):
            g = self.m.contents.regions[r]
            base = 1 + r * stride
            for s in range(4):
                te = (s == 0 and self.model_type == 0)
                num = np.ctypeslib.as_array(g.est[s].num)
                den = np.ctypeslib.as_array(g.est[s].den)
                for p in range(1 if te else 3):
                    nc = 1 if te else ncomp[s]
                    out[base + ((s * 3 + p) * 2 + 0) * K: base + ((s * 3 + p) * 2 + 0) * K + nc] = num[p, :nc]
                    out[base + ((s * 3 + p) * 2 + 1) * K: base + ((s * 3 + p) * 2 + 1) * K + nc] = den[p, :nc]
            cnt = np.ctypeslib.as_array(g.count)
            out[base + 24 * K: base + 24 * K + 16] = cnt[:4, :4].ravel()
        return out

    def labels(self) -> np.ndarray:
        out = []
        for c in range(self.cc.contents.n_chunks):
            ch = self.cc.contents.chunks[c]
            out.append(np.ctypeslib.as_array(ch.prediction, shape=(ch.n,)).copy())
        return np.concatenate(out) if out else np.zeros(0, np.int8)

    def forward_backward(self):
        fs, bs, ss = [], [], []
        for c in range(self.cc.contents.n_chunks):
            ch = self.cc.contents.chunks[c]
            fs.append(np.ctypeslib.as_array(ch.f, shape=(ch.n, 4)).copy())
            bs.append(np.ctypeslib.as_array(ch.b, shape=(ch.n, 4)).copy())
            ss.append(np.ctypeslib.as_array(ch.scales, shape=(ch.n,)).copy())
        return np.concatenate(fs), np.concatenate(bs), np.concatenate(ss)

    def betas(self) -> np.ndarray:
        out = []
        for c in range(self.cc.contents.n_chunks):
            ch = self.cc.contents.chunks[c]
            out.append([self.L.ohf_beta(C.byref(ch), self.cc.contents.window_len, i, C.byref(self.opts))
                        for i in range(ch.n)])
        return np.concatenate(out)

    def estimate_parameters(self, tol: float) -> bool:
        return bool(self.L.ohf_estimate_parameters(self.m, tol))

    def run_em(self, iterations: int, tol: float, out_dir=None, write_params=False, write_posterior=False,
               accelerate=False):
        ll = (C.c_double * (iterations + 2))()
        eo = OEmOpts(iterations, tol, write_params, write_posterior, out_dir.encode() if out_dir else None, accelerate)
        passes = self.L.ohf_run_em(self.cc, self.m, C.byref(self.opts), C.byref(eo), ll, iterations + 2)
        assert passes > 0, f"oracle EM failed: {passes}"
        return list(ll[:passes])

    def write_final_bed(self, path: str, track="final_hmm_flagger", min_len=(0, 0, 0, 0)):
        ml = (C.c_int * 4)(*min_len)
        assert self.L.ohf_write_final_bed(self.cc, path.encode(), track.encode(), ml) == 0
