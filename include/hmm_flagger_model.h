/*
 * hmm_flagger_model.h — C ABI of the host-side model object that sits on top of the E-step:
 * initial parameters, the per-iteration parameter view handed to hf_estep(), and the M-step.
 * These stay on the CPU in the reference's design too — they touch <= a few
 * hundred scalars per iteration.
 *
 * Reference interfaces replaced (mobinasri/flagger, programs/):
 *   createModel                          src/hmm_flagger.c:164-237
 *   HMM_construct                        submodules/hmm/hmm.c:22-77
 *   HMM_estimateParameters               submodules/hmm/hmm.c:120-127
 *   HMM_resetEstimators                  submodules/hmm/hmm.c:129-134 (implicit: the statistics
 *                                        vector of every pass starts from zero)
 *   HMM_printTransitionMatrixInTsvFormat / HMM_printEmissionParametersInTsvFormat   hmm.c:137-239
 *   getBestNumberOfCollapsedComps        src/hmm_flagger.c:105-111, 1012-1013
 *   getAlphaMatrix                       src/hmm_flagger.c:491-515
 */
#ifndef HMM_FLAGGER_MODEL_H
#define HMM_FLAGGER_MODEL_H

#include <stdint.h>
#include "hmm_flagger_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hfm_model hfm_model;

/* alpha16: row-major alpha[preState][state]; NULL = all zero (preset arrays are `int`,
 * src/hmm_flagger.c:21,36,50). Returns NULL on bad arguments. */
hfm_model *hfm_create(int model_type, int n_collapsed, const int32_t *region_coverages, int n_regions,
                      int start_only, int avg_alignment_len, int window_len, const double *alpha16,
                      double max_high_mapq_ratio, double min_high_mapq_ratio);
hfm_model *hfm_copy(const hfm_model *m);
void hfm_destroy(hfm_model *m);

int hfm_n_regions(const hfm_model *m);
int hfm_max_comps(const hfm_model *m);                /* = K of hf_create / the stats layout */
int hfm_model_type(const hfm_model *m);
double hfm_max_high_mapq_ratio(const hfm_model *m);
double hfm_min_high_mapq_ratio(const hfm_model *m);
double hfm_min_highly_clipped_ratio(const hfm_model *m);

/* Parameter view for hf_estep(); pointers stay valid until the model is changed or destroyed. */
void hfm_params(const hfm_model *m, hf_params *out);

/* M-step from one reduced statistics vector (layout: hmm_flagger_hip.h).  Stores stats[0] as the
 * model log-likelihood.  Returns 1 if every parameter moved by less than `tol` (converged), else 0. */
int hfm_estimate(hfm_model *m, const double *stats, double tol);
double hfm_loglikelihood(const hfm_model *m);

int hfm_write_transition_tsv(const hfm_model *m, const char *path);
int hfm_write_emission_tsv(const hfm_model *m, const char *path);

/* flat parameter vector (SQUAREM, tests): [R][ trans 25 | lambda | trunc | mean 4K | var 4K | weight 4K ] */
int64_t hfm_param_len(const hfm_model *m);
void hfm_get_param_vector(const hfm_model *m, double *out);
void hfm_set_param_vector(hfm_model *m, const double *in);

/* --initialRandomDev (hmm_flagger.c:213-220): every initial mean is multiplied by the same random factor
 * (the reference reseeds with time(NULL) at every draw), the collapsed means twice. */
void hfm_scale_initial_means(hfm_model *m, double factor);
/* negative_binomial only: the largest coverage value of the windows the model will meet (<= HF_NB_MAX_COVERAGE); the
 * per-x tables of hfm_params are then filled up to it only.  Default: HF_NB_MAX_COVERAGE. */
void hfm_set_max_coverage(hfm_model *m, int max_x);

/* SQUAREM acceleration (--accelerate): SquareAccelerator_*, submodules/hmm/hmm.c:820-1098.
 * m0 = parameters the last E-step ran with (with its log-likelihood), m1/m2 = after one/two EM updates. */
typedef struct hfm_squarem hfm_squarem;
hfm_squarem *hfm_squarem_create(const hfm_model *m0, const hfm_model *m1, const hfm_model *m2);   /* computeRates :999 */
void hfm_squarem_destroy(hfm_squarem *a);
double hfm_squarem_alpha(const hfm_squarem *a);
/* model prime for the current alpha rate, shrinking alpha until it is feasible (:892-897); owned by `a` */
hfm_model *hfm_squarem_model_prime(hfm_squarem *a);
/* shrink alpha once and again until feasible (:907-910); owned by `a` */
hfm_model *hfm_squarem_shrink(hfm_squarem *a);
int hfm_is_feasible(const hfm_model *m);                                                            /* hmm.c:80-87 */
void hfm_set_loglikelihood(hfm_model *m, double ll);

/* One EM step in one call (E-step on the GPU, statistics to the host, optional M-step): the body of the loop of
 * runHMMFlagger, src/hmm_flagger.c:337-445.  stats_host: [hf_chunk_stats_len]. */
int hf_em_iterate(hf_ctx *ctx, hfm_model *model, int mode, int do_mstep, double tol, double *stats_host, int *converged,
                  void *stream);

/* hf_warmup(device) and then one miniature synthetic EM per kernel family of the default pass (a few milliseconds): the first real pass
 * then finds every kernel launched once.  Optional; for the thread that brings the runtime up while the input is read. */
int hfm_warmup_pipeline(int device);

int hfm_best_collapsed_comps(const uint16_t *cov, int64_t n_windows, const int32_t *region_coverages, int n_regions);
int hfm_read_alpha_tsv(const char *path, double *alpha16);

#ifdef __cplusplus
}
#endif
#endif
