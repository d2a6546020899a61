"""ctypes binding of libhmmflagger_hip.so (include/hmm_flagger_hip.h, include/hmm_flagger_model.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C flagger_amd/csrc`.  There is no
Python or CPU fallback for the E-step: if the shared object is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

HF_NSTATES = 4
HF_MAXCOMP = 16
HF_MAXREGIONS = 64
HF_NKERNELS = 16
HF_MODEL_TRUNC_EXP_GAUSSIAN, HF_MODEL_GAUSSIAN, HF_MODEL_NEGATIVE_BINOMIAL = 0, 1, 2
HF_MODE_FULL, HF_MODE_FORWARD_ONLY = 0, 1
HF_ALGO_SCAN, HF_ALGO_SEQ = 0, 1
HF_STATS_CHUNKS, HF_STATS_ROWS = 0, 1
HF_EXCHANGE_CHUNKS, HF_EXCHANGE_RANKS = 0, 1
HF_TRANSPORT_RCCL, HF_TRANSPORT_LOOPBACK = 0, 1
HF_PROF_PASS = 0x80000000
HF_OK, HF_E_ARG, HF_E_HIP, HF_E_SCALE, HF_E_NAN, HF_E_REGION, HF_E_NOGPU, HF_E_RETRY = 0, -1, -2, -3, -4, -5, -6, -7

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libhmmflagger_hip.so")
# profiling only (profiles/tools/build_variants.sh): HF_LIBRARY_VARIANT=<name> loads csrc/variants/libhmmflagger_hip.<name>.so — the same
# sources built with other -DHF_... switches, for same-box A/B runs without rebuilding on the GPU box
if os.environ.get("HF_LIBRARY_VARIANT"):
    LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "variants", "libhmmflagger_hip.%s.so" % os.environ["HF_LIBRARY_VARIANT"])


class hf_windows(C.Structure):
    _fields_ = [
        ("n_windows", C.c_int64), ("n_chunks", C.c_int32),
        ("chunk_off", C.POINTER(C.c_int64)),
        ("cov", C.POINTER(C.c_uint16)), ("mapq", C.POINTER(C.c_uint16)), ("clip", C.POINTER(C.c_uint16)),
        ("annot", C.POINTER(C.c_uint64)),
        ("chunk_s", C.POINTER(C.c_int32)), ("chunk_e", C.POINTER(C.c_int32)), ("chunk_ctg_len", C.POINTER(C.c_int32)),
        ("window_len", C.c_int32), ("mean_read_len", C.c_int32), ("adjust_contig_ends", C.c_int32),
        ("min_read_frac", C.c_double), ("max_high_mapq_ratio", C.c_double), ("min_high_mapq_ratio", C.c_double),
        ("min_highly_clipped_ratio", C.c_double),
    ]


class hf_params(C.Structure):
    _fields_ = [
        ("model_type", C.c_int32), ("n_regions", C.c_int32), ("ncomp", C.c_int32 * 4),
        ("alpha", (C.c_double * 4) * 4),
        ("trans", C.POINTER(C.c_double)), ("lambda_", C.POINTER(C.c_double)), ("trunc_point", C.POINTER(C.c_double)),
        ("mean", C.POINTER(C.c_double)), ("var", C.POINTER(C.c_double)), ("weight", C.POINTER(C.c_double)),
        ("nb_E", C.POINTER(C.c_double)), ("nb_P", C.POINTER(C.c_double)), ("nb_dig", C.POINTER(C.c_double)),
        ("nb_r", C.POINTER(C.c_double)), ("nb_beta", C.POINTER(C.c_double)),
        ("nb_max_x", C.c_int32),
    ]



class hfs_input(C.Structure):   # include/hmm_flagger_summary.h
    _fields_ = [("n_windows", C.c_int64), ("n_chunks", C.c_int32), ("chunk_off", C.POINTER(C.c_int64)),
                ("chunk_s", C.POINTER(C.c_int32)), ("chunk_e", C.POINTER(C.c_int32)), ("chunk_ctg", C.POINTER(C.c_char_p)),
                ("window_len", C.c_int32), ("annot", C.POINTER(C.c_uint64)), ("truth", C.POINTER(C.c_int8)),
                ("prediction", C.POINTER(C.c_int8)), ("truth_available", C.c_int32), ("prediction_available", C.c_int32),
                ("n_labels", C.c_int32), ("n_regions", C.c_int32), ("n_annotations", C.c_int32),
                ("annotation_names", C.POINTER(C.c_char_p))]

_lib = None


def lib() -> C.CDLL:
    """Load the native library once; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C flagger_amd/csrc`). The HIP E-step has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd = C.POINTER(C.c_double)

    def sig(name, res, *args):
        try:
            fn = getattr(L, name)
        except AttributeError:
            if os.environ.get("HF_LIBRARY_VARIANT"):      # profiling only: a variant built from an older snapshot may lack a newer entry point
                return
            raise
        fn.restype = res
        fn.argtypes = list(args)

    sig("hf_version", C.c_char_p)
    sig("hf_last_error", C.c_char_p)
    sig("hf_device_count", C.c_int)
    sig("hf_warmup", C.c_int, C.c_int)
    sig("hfm_warmup_pipeline", C.c_int, C.c_int)
    sig("hf_create", C.c_int, C.POINTER(hf_windows), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp))
    sig("hf_destroy", None, vp)
    sig("hf_estep", C.c_int, vp, C.POINTER(hf_params), C.c_int, vp)
    sig("hf_n_chunks", i32, vp)
    sig("hf_n_windows", i64, vp)
    sig("hf_chunk_stats_len", i64, vp)
    sig("hf_chunk_stats_dev", vp, vp)
    sig("hf_labels_dev", vp, vp)
    sig("hf_copy_chunk_stats", C.c_int, vp, vp, vp)
    sig("hf_reduce_chunks", C.c_int, vp, vp, i64, vp, vp)
    sig("hf_reduce_chunks_indexed", C.c_int, vp, vp, vp, i64, vp, vp)
    sig("hf_finish", C.c_int, vp, pd, vp)
    sig("hf_finish_gathered", C.c_int, vp, vp, vp, i64, pd, vp)
    sig("hf_check", C.c_int, vp, vp)
    sig("hf_get_labels", C.c_int, vp, C.POINTER(C.c_int8))
    sig("hf_get_posterior", C.c_int, vp, i64, i64, pd)
    sig("hf_get_forward_backward", C.c_int, vp, i64, i64, pd, pd, pd)
    sig("hf_last_kernel_ms", C.c_int, vp, C.POINTER(C.c_float))
    sig("hf_set_profiling", C.c_int, vp, C.c_uint)
    sig("hf_set_profiling_stride", C.c_int, vp, C.c_int)
    sig("hf_kernel_times", C.c_int, vp, C.POINTER(C.c_float))
    sig("hf_kernel_time_sums", C.c_int, vp, pd, C.POINTER(C.c_int64))
    sig("hf_kernel_name", C.c_char_p, C.c_int)
    sig("hf_set_stats_mode", C.c_int, vp, C.c_int)
    sig("hf_rank_total", C.c_int, vp, vp, vp)
    sig("hf_get_stats_mode", C.c_int, vp)
    sig("hf_seg_launches", C.c_int, vp)
    sig("hf_seg_cached_steps", C.c_int, vp)
    sig("hf_sub_passes", C.c_int, vp)
    sig("hf_sub_pass_windows", C.c_int64, vp, C.c_int)
    sig("hf_seg_xcd_plan", C.c_int, vp)
    sig("hf_seg_block_table", C.c_int64, vp, C.POINTER(C.c_int32), C.c_int64)
    sig("hf_create_phases", C.c_int, vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_char_p))
    sig("hf_finish_exchange", C.c_int, vp, vp, vp, i64, C.c_int, C.c_int, C.c_int, pd, vp)
    sig("hf_bind_chunk_stats", C.c_int, vp, vp)
    sig("hf_write_flag_row", C.c_int, vp, vp, vp)
    # multi-GPU (include/hmm_flagger_multi.h)
    sig("hf_comm_unique_id", C.c_int, vp)
    sig("hf_comm_init_rank", C.c_int, C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp))
    sig("hf_comm_init_all", C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(vp))
    sig("hf_comm_init_loopback", C.c_int, C.c_int, C.c_int, C.POINTER(vp))
    sig("hf_comm_destroy", None, vp)
    sig("hf_comm_rank", C.c_int, vp)
    sig("hf_comm_size", C.c_int, vp)
    sig("hf_comm_allgather", C.c_int, vp, vp, vp, i64, vp)
    sig("hf_comm_last_error", C.c_char_p)
    sig("hf_shard_bounds", C.c_int, C.POINTER(i64), i32, C.c_int, C.POINTER(i32))
    sig("hf_multi_create", C.c_int, C.POINTER(hf_windows), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
        C.POINTER(vp))
    sig("hf_multi_create_rank", C.c_int, C.POINTER(hf_windows), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp,
        C.POINTER(vp))
    sig("hf_multi_local_ctx", vp, vp)
    sig("hf_multi_local_first_window", i64, vp)
    sig("hf_multi_local_windows", i64, vp)
    sig("hf_multi_destroy", None, vp)
    sig("hf_multi_estep", C.c_int, vp, C.POINTER(hf_params), C.c_int, pd)
    sig("hf_multi_get_labels", C.c_int, vp, C.POINTER(C.c_int8))
    sig("hf_multi_get_posterior", C.c_int, vp, i64, i64, pd)
    sig("hf_multi_world", C.c_int, vp)
    sig("hf_multi_comm_ranks", C.c_int, vp)
    sig("hf_multi_em_iterate", C.c_int, vp, vp, C.c_int, C.c_int, C.c_double, pd, C.POINTER(C.c_int))
    sig("hf_bind_rank_total", C.c_int, vp, vp, vp)
    sig("hf_multi_stats_len", i64, vp)
    sig("hf_multi_shard_windows", i64, vp, C.c_int)
    sig("hf_multi_shard_chunks", i32, vp, C.c_int)
    sig("hf_multi_rank_stats", C.c_int, vp, C.c_int, pd)
    sig("hf_multi_last_error", C.c_char_p)
    sig("hf_selftest_division", C.c_int, C.c_int, i64, pd, pd, pd, pd, C.POINTER(C.c_int32))
    sig("hf_selftest_exp", C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
    # host model
    sig("hfm_create", vp, C.c_int, C.c_int, C.POINTER(i32), C.c_int, C.c_int, C.c_int, C.c_int, pd, dbl, dbl)
    sig("hfm_copy", vp, vp)
    sig("hfm_destroy", None, vp)
    sig("hfm_n_regions", C.c_int, vp)
    sig("hfm_max_comps", C.c_int, vp)
    sig("hfm_model_type", C.c_int, vp)
    sig("hfm_max_high_mapq_ratio", dbl, vp)
    sig("hfm_min_high_mapq_ratio", dbl, vp)
    sig("hfm_min_highly_clipped_ratio", dbl, vp)
    sig("hfm_params", None, vp, C.POINTER(hf_params))
    sig("hfm_estimate", C.c_int, vp, pd, dbl)
    sig("hfm_loglikelihood", dbl, vp)
    sig("hfm_write_transition_tsv", C.c_int, vp, C.c_char_p)
    sig("hfm_write_emission_tsv", C.c_int, vp, C.c_char_p)
    sig("hfm_param_len", i64, vp)
    sig("hfm_get_param_vector", None, vp, pd)
    sig("hfm_set_param_vector", None, vp, pd)
    sig("hfm_best_collapsed_comps", C.c_int, C.POINTER(C.c_uint16), i64, C.POINTER(i32), C.c_int)
    sig("hfm_read_alpha_tsv", C.c_int, C.c_char_p, pd)
    sig("hf_em_iterate", C.c_int, vp, vp, C.c_int, C.c_int, dbl, pd, C.POINTER(C.c_int), vp)
    # summary tables (include/hmm_flagger_summary.h)
    sig("hfs_write_all_tables", C.c_int, C.POINTER(hfs_input), C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, dbl, C.c_int)
    sig("hfs_last_error", C.c_char_p)
    # SQUAREM + misc model helpers
    sig("hfm_scale_initial_means", None, vp, dbl)
    sig("hfm_set_max_coverage", None, vp, C.c_int)
    sig("hfm_squarem_create", vp, vp, vp, vp)
    sig("hfm_squarem_destroy", None, vp)
    sig("hfm_squarem_alpha", dbl, vp)
    sig("hfm_squarem_model_prime", vp, vp)
    sig("hfm_squarem_shrink", vp, vp)
    sig("hfm_is_feasible", C.c_int, vp)
    sig("hfm_set_loglikelihood", None, vp, dbl)
    # window table / file formats
    sig("hfio_load", vp, C.c_char_p, C.c_int, C.c_int)
    sig("hfio_destroy", None, vp)
    sig("hfio_last_error", C.c_char_p)
    sig("hfio_n_windows", i64, vp)
    sig("hfio_n_chunks", i32, vp)
    sig("hfio_n_regions", i32, vp)
    sig("hfio_region_coverages", C.POINTER(i32), vp)
    sig("hfio_window_len", i32, vp)
    sig("hfio_chunk_len", i32, vp)
    sig("hfio_avg_alignment_len", i32, vp)
    sig("hfio_start_only", i32, vp)
    sig("hfio_n_annotations", i32, vp)
    sig("hfio_annotation_name", C.c_char_p, vp, C.c_int)
    sig("hfio_chunk_ctg", C.c_char_p, vp, C.c_int)
    sig("hfio_truth", C.POINTER(C.c_int8), vp)
    sig("hfio_prediction", C.POINTER(C.c_int8), vp)
    sig("hfio_windows", None, vp, C.POINTER(hf_windows))
    sig("hfio_write_bin", C.c_int, vp, C.c_char_p)
    sig("hfio_write_final_bed", C.c_int, vp, C.POINTER(C.c_int8), C.c_char_p, C.c_char_p, C.POINTER(i32))
    sig("hfio_write_posterior_bed", C.c_int, vp, pd, C.POINTER(C.c_int8), C.c_char_p)
    _lib = L
    return L


def region_stride(max_comps: int) -> int:
    return 24 * max_comps + 16


def stats_len(n_regions: int, max_comps: int) -> int:
    return 1 + n_regions * region_stride(max_comps)


class HFError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        msg = lib().hf_last_error().decode(errors="replace")
        super().__init__(f"{where} failed with code {code}: {msg}")


def check(code: int, where: str) -> None:
    if code != HF_OK:
        raise HFError(code, where)
