/*
 * hmm_flagger_hip.h — C ABI of the MI355X (gfx950) E-step of HMM-Flagger.
 *
 * This is the drop-in boundary.  The reference has no FFI; the seam is the C
 * function pair a maintainer would re-point at this library (citations relative to
 * mobinasri/flagger, programs/submodules/hmm/):
 *
 *   void EM_runOneIterationForList(stList *emList, HMM *model, int threads);  hmm.h:109, hmm.c:739
 *   void EM_runForwardForList   (stList *emList, HMM *model, int threads);  hmm.h:113, hmm.c:790
 *   double *EM_getPosterior(EM *em, int pos);                               hmm.h:101, hmm.c:671
 *   int     EM_getMostProbableState(EM *em, int pos);                       hmm.h:103, hmm.c:687
 *
 * Plain pointers and sizes only.  Window arrays are uploaded once (hf_create); per iteration
 * only the parameter block goes up and the sufficient statistics + labels come back.
 * All functions return 0 on success or a negative HF_E_* code; where the reference calls
 * exit(EXIT_FAILURE) (hmm.c:412-415, 521-524; hmm_utils.c:782-786) the code says which check
 * fired and the caller maps it to the reference's message and exit status.
 */
#ifndef HMM_FLAGGER_HIP_H
#define HMM_FLAGGER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HF_NSTATES 4      /* Err, Dup, Hap, Col — hmm_flagger.c:173 */
#define HF_MAXCOMP 16     /* per-state mixture components (reference clamps K to 2..10) */
#define HF_MAXREGIONS 64  /* 6 region bits — ptBlock.c:294-304 */

enum { HF_MODEL_TRUNC_EXP_GAUSSIAN = 0, HF_MODEL_GAUSSIAN = 1, HF_MODEL_NEGATIVE_BINOMIAL = 2 };   /* hmm_utils.h:43-48 */
#define HF_NB_MAX_COVERAGE 250   /* MAX_COVERAGE_VALUE, hmm_utils.h:15: count data has bins 0..249, tables 0..250 */
enum { HF_MODE_FULL = 0,          /* EM_runOneIterationForList */
       HF_MODE_FORWARD_ONLY = 1   /* EM_runForwardForList */ };
enum { HF_ALGO_SCAN = 0,          /* in-chunk parallel prefix scan (default) */
       HF_ALGO_SEQ = 1            /* one wavefront per chunk, reference operation order */ };

enum {
    HF_OK = 0,
    HF_E_ARG = -1,        /* bad argument */
    HF_E_HIP = -2,        /* HIP runtime error (hf_last_error() has the text) */
    HF_E_SCALE = -3,      /* "scale (= ...) is very low!"      hmm.c:412-415, 521-524 */
    HF_E_NAN = -4,        /* "[Error] prob is NAN"             hmm_utils.c:782-786 */
    HF_E_REGION = -5,     /* a window's region index >= n_regions */
    HF_E_NOGPU = -6,      /* no HIP device: there is no CPU fallback */
    HF_E_RETRY = -7       /* hf_finish_exchange / hf_finish_gathered / hf_check only: the context changed its launch mode (a hand-off
                           * inside the one-launch segment kernel timed out) and the pass has to be run again: hf_estep, exchange,
                           * finish — with an exchange every rank gets this code together.  hf_finish and hf_em_iterate re-run the
                           * pass themselves. */
};

typedef struct hf_ctx hf_ctx;

/* The windowed coverage track, in the reference's packed-record fields (chunk.c:669-697):
 * one entry per window, chunks delimited by chunk_off; all arrays in HOST memory. */
typedef struct hf_windows {
    int64_t n_windows;
    int32_t n_chunks;
    const int64_t *chunk_off;        /* [n_chunks+1] first window of each chunk */
    const uint16_t *cov;             /* CoverageInfo.coverage            ptBlock.h:79-92 */
    const uint16_t *mapq;            /* CoverageInfo.coverage_high_mapq */
    const uint16_t *clip;            /* CoverageInfo.coverage_high_clip */
    const uint64_t *annot;           /* annotation_flag, region index in bits 58..63 */
    const int32_t *chunk_s;          /* [n_chunks] Chunk.s (0-based)     chunk.h:16-23 */
    const int32_t *chunk_e;          /* [n_chunks] Chunk.e (0-based, inclusive) */
    const int32_t *chunk_ctg_len;    /* [n_chunks] Chunk.ctgLen */
    int32_t window_len;              /* Chunk.windowLen */
    int32_t mean_read_len;           /* header averageAlignmentLength    hmm_flagger.c:312 */
    int32_t adjust_contig_ends;      /* !--disableAdjustContigEnds       hmm_flagger.c:624 */
    double min_read_frac;            /* --minReadFractionAtEnds          hmm.c:305 */
    double max_high_mapq_ratio;      /* TransitionRequirements           hmm_utils.c:1950-1958 */
    double min_high_mapq_ratio;
    double min_highly_clipped_ratio;
} hf_windows;

/* One iteration's model (HMM struct, hmm.h:14-25), HOST memory, read during hf_estep only. */
typedef struct hf_params {
    int32_t model_type;              /* HF_MODEL_* */
    int32_t n_regions;
    int32_t ncomp[HF_NSTATES];       /* components per state */
    double alpha[4][4];              /* alpha[preState][state]           hmm.c:388 */
    const double *trans;             /* [n_regions][5][5] row 4 = Start, col 4 = End */
    const double *lambda;            /* [n_regions] Err trunc-exp rate   hmm_utils.h:393-397 */
    const double *trunc_point;       /* [n_regions] */
    const double *mean;              /* [n_regions][4][HF_MAXCOMP] */
    const double *var;               /* [n_regions][4][HF_MAXCOMP] */
    const double *weight;            /* [n_regions][4][HF_MAXCOMP] */
    /* HF_MODEL_NEGATIVE_BINOMIAL only (NULL otherwise); mean = theta, var = lambda of the NegativeBinomial struct.
     * Every quantity of that model depends on the coverage x alone (no alpha, no beta), so the caller tabulates it
     * with its own libm, K = the max_comps given to hf_create, NX = HF_NB_MAX_COVERAGE + 1:
     *   nb_E[r][s][x]        emission value, NegativeBinomial_getProb                      hmm_utils.c:480-485
     *   nb_P[r][s][c][x]     component probabilities (weights and 1e-40 floor applied)     hmm_utils.c:497-520
     *   nb_dig[r][s][c][x]   NegativeBinomial.digammaTable                                 hmm_utils.c:394-408
     *   nb_r, nb_beta[r][s][c]   r = -lambda/log(theta), beta = -theta/(1-theta) - 1/log(theta)   :458-461, 547 */
    const double *nb_E, *nb_P, *nb_dig, *nb_r, *nb_beta;
    /* <= 0: the nb_* tables are filled for every x; else the largest x for which they are (hfm_set_max_coverage): hf_estep
     * refuses a context whose windows exceed it.  The tables cost ~70 ns per (component, x) on the host every iteration. */
    int32_t nb_max_x;
} hf_params;

/* Layout of one statistics vector (doubles):
 *   [0]                                   log-likelihood (sum of log scale, hmm.c:428)
 *   per region r, base = 1 + r*hf_region_stride(K):
 *     base + ((s*3 + p)*2 + 0)*K + c      numeratorPerComp[c]   of state s, parameter p
 *     base + ((s*3 + p)*2 + 1)*K + c      denominatorPerComp[c]
 *         p = 0 mean (or trunc-exp lambda for Err), 1 var, 2 weight   hmm_utils.h:50-54,64-68
 *     base + 24*K + pre*4 + s             TransitionCountData.countMatrix[pre][s]
 * with K = max_comps given to hf_create. */
static inline int64_t hf_region_stride(int max_comps) { return 24 * (int64_t) max_comps + 16; }
static inline int64_t hf_stats_len(int n_regions, int max_comps) {
    return 1 + (int64_t) n_regions * hf_region_stride(max_comps);
}

const char *hf_version(void);
const char *hf_last_error(void);
int hf_device_count(void);
/* Optional: bring up the HIP runtime and the context of `device` ahead of hf_create (≈ 0.1 s), e.g. on another thread
 * while the input is being read.  Returns HF_E_NOGPU without a device. */
int hf_warmup(int device);

/* Upload the windows to `device` and build the device-resident window store. */
int hf_create(const hf_windows *w, int n_regions, int max_comps, int device, int algo, hf_ctx **out);
void hf_destroy(hf_ctx *ctx);

/* Launch one E-step pass over every chunk of this context on `stream` (a hipStream_t, NULL =
 * default stream); asynchronous.  Leaves one statistics vector PER CHUNK on the device. */
int hf_estep(hf_ctx *ctx, const hf_params *p, int mode, void *stream);

int32_t hf_n_chunks(const hf_ctx *ctx);
int64_t hf_n_windows(const hf_ctx *ctx);
int64_t hf_chunk_stats_len(const hf_ctx *ctx);        /* = hf_stats_len(n_regions, max_comps) */
double *hf_chunk_stats_dev(hf_ctx *ctx);              /* device [n_chunks][hf_chunk_stats_len] */
int8_t *hf_labels_dev(hf_ctx *ctx);                   /* device [n_windows] */
/* device-to-device copy of the per-chunk vectors into a caller-owned device buffer (e.g. the
 * send buffer of the multi-GPU all-gather); asynchronous on `stream`. */
int hf_copy_chunk_stats(hf_ctx *ctx, double *dst_dev, void *stream);

/* Sum `n_chunks` per-chunk vectors (the reference's reduction over the chunk list, hmm.c:759-763) into one
 * vector, in a fixed order that depends only on the position of a chunk in the list; src/dst are DEVICE
 * pointers; asynchronous on `stream`. */
int hf_reduce_chunks(hf_ctx *ctx, const double *chunk_stats_dev, int64_t n_chunks, double *out_dev, void *stream);
/* Same, with the vector of list position c taken from row row_index_dev[c] of `rows_dev` (multi-GPU: the
 * all-gathered buffer holds every rank's rows padded to a common count); identical result to the packed form. */
int hf_reduce_chunks_indexed(hf_ctx *ctx, const double *rows_dev, const int32_t *row_index_dev, int64_t n_chunks,
                             double *out_dev, void *stream);

/* Multi-GPU counterpart of hf_finish: `rows_dev` holds the per-chunk vectors of ALL ranks (all-gathered; row_index_dev[c]
 * = row of list position c, or NULL when packed), summed in the fixed order into `stats_host`; waits for the stream
 * and translates the device error flags of this rank's pass.  Every rank gets identical bits. */
int hf_finish_gathered(hf_ctx *ctx, const double *rows_dev, const int32_t *row_index_dev, int64_t n_chunks,
                       double *stats_host, void *stream);

/* The same with the error flags of EVERY rank: the gathered buffer holds `rows_per_rank` rows per rank and row
 * `flag_row` of each rank carries that rank's device flag word (hf_write_flag_row); the words are OR-ed into the result,
 * so all ranks return the same HF_E_* code and none is left waiting in the next collective (hmm.c:412-415 exits the whole
 * process; here the whole job stops).  rows_per_rank >= 2, 0 <= flag_row < rows_per_rank. */
int hf_finish_exchange(hf_ctx *ctx, const double *rows_dev, const int32_t *row_index_dev, int64_t n_rows, int n_ranks,
                       int rows_per_rank, int flag_row, double *stats_host, void *stream);
/* Let the pass write its per-chunk vectors straight into caller-owned device memory (>= n_chunks rows), e.g. this rank's
 * slot of an in-place all-gather buffer: no device-to-device copy per pass.  NULL: back to a buffer of the context. */
int hf_bind_chunk_stats(hf_ctx *ctx, double *rows_dev);
/* The same for the `ranks` exchange: a full pass in HF_STATS_ROWS mode (Gaussian / trunc-exp models) writes its total (V
 * doubles) straight to `total_dev` and its error-flag word to element 0 of `flag_row_dev`; hf_rank_total(total_dev) and
 * hf_write_flag_row(flag_row_dev) after such a pass then launch nothing.  Passes that do not produce a total of their own
 * (forward-only, negative_binomial, per-chunk statistics) are unaffected.  NULL, NULL: unbind. */
int hf_bind_rank_total(hf_ctx *ctx, double *total_dev, double *flag_row_dev);
/* This context's device error-flag word as element 0 of `row_dev` (device memory, one row of the exchange buffer);
 * asynchronous on `stream`, after the pass. */
int hf_write_flag_row(hf_ctx *ctx, double *row_dev, void *stream);

/* Single-GPU convenience: reduce this context's chunks, copy the vector to `stats_host`,
 * wait for the stream and translate the device error flags (HF_E_SCALE / HF_E_NAN / ...). */
/* (hf_finish synchronises the stream.  Environment HF_POLL=1 opts into polling a checksummed completion stamp in the
 * pinned result block instead — a few microseconds less per pass, see hf_estep.hip — HF_POLL=debug verifies it.) */
int hf_finish(hf_ctx *ctx, double *stats_host, void *stream);
/* Only wait + error flags (multi-GPU callers reduce the gathered vectors themselves).  HF_E_RETRY: a hand-off of the one-launch
 * segment kernel timed out; the context has switched to two launches and the caller repeats the pass from hf_estep. */
int hf_check(hf_ctx *ctx, void *stream);

/* How a HF_MODE_FULL pass of HF_ALGO_SCAN produces the statistics:
 *   HF_STATS_CHUNKS  one estimator vector per chunk (EM_runOneIterationForList's per-chunk EM objects, hmm.c:739-763),
 *                    reduced over the chunk list in list order: hf_chunk_stats_dev / hf_copy_chunk_stats /
 *                    hf_reduce_chunks* / hf_finish_gathered work on these vectors, and the result does not depend on how
 *                    the chunk list is sharded over GPUs (bit for bit).
 *   HF_STATS_ROWS    the pair counts are summed per emission row first and the estimator updates run once per row
 *                    (flagger_amd/csrc/hf_rows.h); hf_finish returns the same vector up to the rounding of a different
 *                    summation order (fixed by the plan of hf_create: reproducible), ~2.5x less statistics time.  The
 *                    per-chunk vectors are NOT produced (only element 0, the chunk's log-likelihood).
 * Default: HF_STATS_ROWS where it applies — HF_ALGO_SCAN and a plan that is not
 * dominated by padding (it is when nearly every window has a private emission row: reads longer than the contigs) —
 * else HF_STATS_CHUNKS is used silently (hf_get_stats_mode tells); environment HF_STATS=chunks|rows
 * overrides the default at hf_create. */
enum { HF_STATS_CHUNKS = 0, HF_STATS_ROWS = 1 };
int hf_set_stats_mode(hf_ctx *ctx, int mode);
/* The statistics vector of THIS context's chunks after a pass, in either mode, into device memory (hf_chunk_stats_len
 * doubles): what ranks exchange when the per-chunk vectors are not needed — all-gather one vector per rank, then
 * hf_finish_gathered(rows = the gathered vectors, row_index = NULL, n = world size) sums them in rank order. */
int hf_rank_total(hf_ctx *ctx, double *out_dev, void *stream);
int hf_get_stats_mode(const hf_ctx *ctx);          /* the mode the NEXT full pass will use */
/* Launches of the segment forward-backward in the NEXT pass: 1 = k_seg_fb alone (its segments hand their products over inside the
 * launch), 2 = k_seg_prod + k_seg_fb (hf_create's choice for a chunk with more segments than the device holds workgroups,
 * environment HF_SEG_LAUNCHES=2, or after a hand-off timed out), 0 = the context does not run the segment kernels (HF_ALGO_SEQ, empty). */
int hf_seg_launches(const hf_ctx *ctx);
/* Row blocks (of at most 8 = windows per lane) that a segment workgroup of the NEXT one-launch pass keeps in LDS across its three walks
 * instead of fetching them again: hf_create's choice — all 8 when every segment is still resident together with that much LDS each (up to
 * ~430 segments on 256 CUs: a 1/8 shard of BASELINE configs[2]), 0 otherwise (environment HF_SEG_CACHED_STEPS forces any number). */
int hf_seg_cached_steps(const hf_ctx *ctx);
/* Sub-passes of a full pass: a context whose pair records (64 bytes per window) would not fit the 256 MB Infinity Cache — more than ~2.8 M
 * windows on one GPU — cuts its chunk list into sub-passes of whole chunks (<= ~1.6 M windows each) and runs the segment kernel and the
 * per-group sums sub-pass by sub-pass through one record buffer; 1 for BASELINE configs[2] (environment HF_SUBPASSES forces a number).
 * hf_sub_pass_windows: the windows of sub-pass k (what ONE launch of the segment kernel processes: hf_set_profiling times the first). */
int hf_sub_passes(const hf_ctx *ctx);
int64_t hf_sub_pass_windows(const hf_ctx *ctx, int k);
/* XCD plan of the one-launch segment kernel (round 6): 1 when the blocks of the NEXT pass run the segments through hf_create's block ->
 * segment table (all segments of a chunk on block indices congruent mod 8 — observed: one XCD; the hand-off stays system-scope, so this is
 * for speed only), 0 when block b runs segment b (environment HF_SEG_XCD=0, two-launch mode, or a chunk whose segments would span more
 * than half of the resident workgroups).  hf_seg_block_table: the table (n entries copied, -1 = padding block); returns its length. */
int hf_seg_xcd_plan(const hf_ctx *ctx);
int64_t hf_seg_block_table(const hf_ctx *ctx, int32_t *seg_of_block, int64_t n);
/* Where hf_create's wall time went (what EM_construct + EM_renewParametersAndEstimatorsFromModel cost the reference per chunk and
 * iteration, hmm.c:253-298, paid once here): up to `max` phases in call order, ms[i] and a static name each; returns the number of
 * phases recorded.  The last entry is the total. */
int hf_create_phases(const hf_ctx *ctx, int max, double *ms, const char **names);

/* Results of the last HF_MODE_FULL pass (HF_E_ARG when the last pass was HF_MODE_FORWARD_ONLY: f and scales would be new,
 * b and the labels stale).  hf_get_posterior / hf_get_forward_backward: an EM pass of the default algorithm keeps no per-window scale (and,
 * in several sub-passes, only the last sub-pass's records) — the first call after a pass runs the segment kernel once more over the pass's
 * tables, with the scale array (~0.05 ms per 1.5 M windows; the array itself is allocated by that first call); same values as the pass. */
int hf_get_labels(hf_ctx *ctx, int8_t *labels_host);                                   /* hmm.c:730-736 */
int hf_get_posterior(hf_ctx *ctx, int64_t first, int64_t n, double *post_host);        /* [n][4] hmm.c:671-685 */
int hf_get_forward_backward(hf_ctx *ctx, int64_t first, int64_t n, double *f_host, double *b_host,
                            double *scales_host);                                      /* EM.f/.b/.scales */
/* kernel time of the last hf_estep + reduce in milliseconds (HIP events on the stream used); recorded only while
 * hf_set_profiling's mask carries HF_PROF_PASS (two extra stream packets per pass) */
int hf_last_kernel_ms(hf_ctx *ctx, float *ms);

/* Per-kernel timing (bench.py's roofline leg): every kernel k whose bit is set in kernel_mask is bracketed
 * by a pair of HIP events on the launch stream; hf_kernel_times returns the duration of each selected kernel in
 * the LAST pass in milliseconds (0 for kernels not selected or not run).  Call after hf_finish/hf_check.
 * Each selected kernel adds two event packets to the stream, so select only what is being measured. */
#define HF_NKERNELS 16
/* indices 1..3 are reserved (round 1's tile kernels, retired in round 2: no name, never run; the later indices keep their values) */
enum { HF_K_TABLES = 0, HF_K_STATS_TILE = 4, HF_K_CHUNK_STATS, HF_K_REDUCE,
       HF_K_EMIT_ROWS, HF_K_FWD_SEQ, HF_K_BWD_SEQ, HF_K_PAIR_SUMS, HF_K_ROW_STATS, HF_K_ROWS_TOTAL, HF_K_SEG_PROD, HF_K_SEG_FB, HF_K_AROWS };
#define HF_PROF_PASS 0x80000000u   /* in kernel_mask: also bracket the whole pass (hf_last_kernel_ms) */
int hf_set_profiling(hf_ctx *ctx, unsigned kernel_mask);
/* Bracket the selected kernels only in every n-th pass (default 1: every pass): an event pair costs a few microseconds of
 * a 0.2 ms pass, so a timed loop can sample the dominant kernel's duration without carrying the cost in every step. */
int hf_set_profiling_stride(hf_ctx *ctx, int every_nth_pass);
int hf_kernel_times(hf_ctx *ctx, float ms[HF_NKERNELS]);
/* Sum of the durations (ms) and number of timed launches of every selected kernel over all passes finished by
 * hf_finish / hf_em_iterate since the last hf_set_profiling: one call after a timed loop instead of one per pass. */
int hf_kernel_time_sums(hf_ctx *ctx, double sum_ms[HF_NKERNELS], int64_t launches[HF_NKERNELS]);
const char *hf_kernel_name(int k);   /* "k_tables", "k_seg_fb", ... as they appear in a rocprofv3 kernel trace */

/* Self-test hook: fast[i] = a[i] / d[i] through the shared-denominator form the statistics kernel uses (hf_device.h
 * prediv / divp), exact[i] = the plain division, safe[i] = whether the kernel's guard would take the fast form.
 * fast must equal exact bit for bit wherever safe is set. */
int hf_selftest_division(int device, int64_t n, const double *a, const double *d, double *fast, double *exact, int32_t *safe);
/* Self-test hook: csrc/hf_exp.h — glibc's exp for double restated, fused where the host's FMA build fuses (the emission densities of the
 * reference call libm's: hmm_utils.c:782, 945) — for n arguments: dev_out[i] computed on the device, host_out[i] by the same function compiled
 * for the host, libm_out[i] = the host's libm exp.  All three must hold the same bits on a host whose libm runs its FMA variant.  (The
 * emission kernels themselves use the device library's exp unless built with -DHF_EXP_OCML=0: hf_device.h says why.) */
int hf_selftest_exp(int device, int64_t n, const double *x, double *dev_out, double *host_out, double *libm_out);

#ifdef __cplusplus
}
#endif
#endif
