/*
 * ohf.h — ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C CPU restatement of HMM-Flagger's EM hot path (emission evaluation, scaled
 * forward/backward, posterior decode, EM sufficient statistics, M-step, EM outer loop).
 * Every function cites the reference file:line (relative to /root/reference/) it follows.
 *
 * PARITY UNPINNED by the reference's own test-suite: the reference has no automated test,
 * golden vector or fixture for the EM path, and the reference C sources
 * cannot be compiled here without stand-ins for sonLib/htslib (absent from this image), so no
 * oracle/_ref build exists.  Soft pins that ARE checked (tests/test_oracle_*.py):
 *   - docs/hmm_test/README.md recipe: parameters recovered from data simulated by the
 *     reference's own programs/src/simulate_coverage_data.py within its rel.diff<0.1 criterion;
 *   - the loader's window averaging against programs/tests/test_chunks_creator.c expectations.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this code.
 */
#ifndef OHF_H
#define OHF_H

#include <stdint.h>
#include <stdbool.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OHF_NSTATES 4            /* Err, Dup, Hap, Col — hmm_flagger.c:173 */
#define OHF_MAXCOMP 16           /* reference auto-clamps K to [2,10] (hmm_flagger.c:1012-1013) */
#define OHF_MAXREGIONS 64        /* 6 region bits — ptBlock.c:294-304 */
#define OHF_PI 3.14159           /* common.h:15 (sic) */
#define OHF_TERMINATION_PROB 1e-4 /* hmm_utils.c:2112 */

#define OHF_MAX_COVERAGE_VALUE 250 /* hmm_utils.h:15 */

enum { OHF_MODEL_TRUNC_EXP_GAUSSIAN = 0, OHF_MODEL_GAUSSIAN = 1, OHF_MODEL_NEGATIVE_BINOMIAL = 2 }; /* hmm_utils.h:43-48 */
enum { OHF_STATE_ERR = 0, OHF_STATE_DUP = 1, OHF_STATE_HAP = 2, OHF_STATE_COL = 3 };
enum { OHF_P_MEAN = 0, OHF_P_VAR = 1, OHF_P_WEIGHT = 2 };          /* hmm_utils.h:50-54 */
enum { OHF_P_NB_THETA = 0, OHF_P_NB_LAMBDA = 1 };                  /* hmm_utils.h:56-62 (weight = 2 as well) */

/* numerator/denominator accumulators of one emission distribution (hmm_utils.h:93-107);
 * trunc-exp uses [0][0] only (lambda estimator, hmm_utils.c:1310). */
typedef struct {
    double num[3][OHF_MAXCOMP];
    double den[3][OHF_MAXCOMP];
} ohf_estimator;

typedef struct {
    /* transition: 5x5, row 4 = Start, column 4 = End (hmm_utils.c:2109-2128) */
    double trans[5][5];
    double pseudo[5][5];
    double count[5][5];
    /* Err as truncated exponential (trunc_exp_gaussian model) */
    double lambda, trunc_point;
    /* Gaussian mixtures; state Err only in the gaussian model */
    double mean[OHF_NSTATES][OHF_MAXCOMP];
    double var[OHF_NSTATES][OHF_MAXCOMP];
    double weight[OHF_NSTATES][OHF_MAXCOMP];
    /* negative_binomial model: theta, lambda per component (weights above), hmm_utils.h NegativeBinomial */
    double theta[OHF_NSTATES][OHF_MAXCOMP];
    double nb_lambda[OHF_NSTATES][OHF_MAXCOMP];
    ohf_estimator est[OHF_NSTATES];
} ohf_region;

typedef struct {
    int model_type;
    int n_regions;
    int ncomp[OHF_NSTATES];
    double alpha[4][4];            /* alpha[pre][state] — hmm.c:388 */
    double max_high_mapq_ratio;    /* hmm_flagger.c:625 */
    double min_high_mapq_ratio;    /* hmm_flagger.c:626 */
    double min_highly_clipped_ratio; /* hmm_flagger.c:222 */
    double loglikelihood;
    ohf_region *regions;
} ohf_model;

typedef struct {
    char ctg[200];
    int32_t ctg_len, s, e;         /* 0-based inclusive — chunk.h:16-23 */
    int32_t n;                     /* coverageInfoSeqLen */
    uint16_t *cov, *mapq, *clip;   /* window averages, <= 250 — chunk.c:410-415 */
    uint64_t *annot;               /* region in bits 58..63 — ptBlock.c:294-304 */
    int8_t *truth, *prediction;
    /* filled by the E-step when requested */
    double *f, *b, *scales;        /* [n*4], [n*4], [n] */
    double loglikelihood;
} ohf_chunk;

typedef struct {
    int32_t n_annotations;
    char **annotation_names;
    int32_t n_regions;
    int32_t region_coverages[OHF_MAXREGIONS];
    int32_t n_labels;
    bool truth_available, prediction_available, start_only;
    int32_t avg_alignment_len;
    int32_t chunk_len, window_len;
    int32_t n_chunks;
    ohf_chunk *chunks;
} ohf_chunks;

typedef struct {
    bool adjust_contig_ends;       /* hmm_flagger.c:624 */
    double min_read_frac;          /* hmm_flagger.c:640,948-950 */
    int mean_read_len;             /* hmm_flagger.c:312 */
    int threads;
} ohf_run_opts;

/* ---- E-step (ohf_estep.c) ---- */
double ohf_beta(const ohf_chunk *ch, int window_len, int col, const ohf_run_opts *o);
double ohf_emission(const ohf_model *m, int region, int state, uint8_t x, uint8_t pre_x,
                    double alpha, double beta, int *err);
double ohf_trans_cond(const ohf_model *m, int region, int pre, int state,
                      uint16_t cov, uint16_t mapq, uint16_t clip);
/* one chunk; est/count of `acc` (n_regions regions) are incremented; returns 0 or <0 on the
 * reference's exit(EXIT_FAILURE) conditions */
int ohf_chunk_forward(ohf_chunk *ch, const ohf_model *m, int window_len, const ohf_run_opts *o);
int ohf_chunk_backward(ohf_chunk *ch, const ohf_model *m, int window_len, const ohf_run_opts *o);
int ohf_chunk_update_estimators(ohf_chunk *ch, const ohf_model *m, ohf_region *acc, int window_len,
                                const ohf_run_opts *o);
void ohf_posterior(const ohf_chunk *ch, int pos, double post[4]);
int ohf_most_probable_state(const ohf_chunk *ch, int pos);
/* EM_runOneIterationForList (forward_only=0) / EM_runForwardForList (forward_only=1) */
int ohf_run_iteration(ohf_chunks *cc, ohf_model *m, const ohf_run_opts *o, int forward_only);

/* ---- model + M-step (ohf_model.c) ---- */
ohf_model *ohf_model_create(int model_type, int n_collapsed, const int32_t *region_coverages,
                            int n_regions, bool start_only, int avg_alignment_len, int window_len,
                            const double alpha[4][4], double max_high_mapq_ratio,
                            double min_high_mapq_ratio);
void ohf_model_destroy(ohf_model *m);
bool ohf_estimate_parameters(ohf_model *m, double tol);   /* HMM_estimateParameters */
void ohf_reset_estimators(ohf_model *m);                  /* HMM_resetEstimators */
int ohf_best_collapsed_comps(const ohf_chunks *cc);       /* hmm_flagger.c:105-111,1012-1013 */
double ohf_estimate_lambda(double trunc_point, double num, double den, double tol);

/* ---- negative_binomial model (ohf_nb.c) ---- */
long double ohf_digammal(long double x);
double ohf_nb_r(double theta, double lambda);
double ohf_nb_mean(double theta, double lambda);
double ohf_nb_var(double theta, double lambda);
void ohf_nb_init(ohf_region *g, int s, const double *mean, int ncomp);
int ohf_nb_comp_probs(const ohf_region *g, int s, int ncomp, uint8_t x, double *probs);
void ohf_nb_digamma_table(const ohf_region *g, int s, int ncomp, double table[][OHF_MAX_COVERAGE_VALUE + 1]);
int ohf_nb_update(ohf_estimator *est, const ohf_region *g, int s, int ncomp,
                  double table[][OHF_MAX_COVERAGE_VALUE + 1], uint8_t x, double count);
int ohf_nb_update_from_counts(ohf_region *acc, const ohf_region *g, const int *ncomp,
                              double counts[OHF_NSTATES][OHF_MAX_COVERAGE_VALUE]);
bool ohf_nb_estimate(const ohf_model *m, ohf_region *g, double tol);

/* ---- I/O (ohf_io.c) ---- */
ohf_chunks *ohf_read_bin(const char *path);
int ohf_write_bin(const ohf_chunks *cc, const char *path);
void ohf_chunks_destroy(ohf_chunks *cc);
int ohf_read_alpha_tsv(const char *path, double alpha[4][4]);
void ohf_write_transition_tsv(const ohf_model *m, FILE *f);
void ohf_write_emission_tsv(const ohf_model *m, FILE *f);
int ohf_write_final_bed(const ohf_chunks *cc, const char *path, const char *track_name,
                        const int min_len_per_state[4]);
int ohf_write_posterior_bed(const ohf_chunks *cc, const char *path);

/* ---- whole run (ohf_run.c): runHMMFlagger, hmm_flagger.c:285-488 ---- */
typedef struct {
    int iterations;
    double tol;
    bool write_params_per_iter, write_posterior;
    const char *out_dir;           /* may be NULL: no files */
    bool accelerate;               /* --accelerate (SQUAREM), hmm_flagger.c:382-416 */
} ohf_em_opts;
int ohf_squarem_iteration(ohf_chunks *cc, ohf_model *m, const ohf_run_opts *o, double tol, double *alpha_out);
/* returns number of E-passes executed (I+1) or <0 */
int ohf_run_em(ohf_chunks *cc, ohf_model *m, const ohf_run_opts *o, const ohf_em_opts *eo,
               double *ll_trace, int ll_cap);

#ifdef __cplusplus
}
#endif
#endif
