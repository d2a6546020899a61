/*
 * ohf_estep.c — ORACLE (test infrastructure, not product code).
 * CPU restatement of the per-chunk E-step of HMM-Flagger, same fp64 operation order as the
 * reference.  Citations are file:line under /root/reference/programs/submodules/.
 * Compile with -ffp-contract=off -fno-builtin-pow (the shipped reference is built without -O,
 * programs/Makefile:4, so pow(d,2) is the libm call, not d*d).
 */
#include "ohf.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* common.c:142-148 — min/max take int: every argument is truncated toward zero at the call */
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return b < a ? a : b; }

static inline int region_of(uint64_t annot) { /* ptBlock.c:294-298 */
    return (int) ((annot & 0xFC00000000000000ULL) >> (64 - 6));
}

/* hmm.c:301-316 EM_computeAdjustmentBeta */
double ohf_beta(const ohf_chunk *ch, int window_len, int col, const ohf_run_opts *o) {
    if (!o->adjust_contig_ends) return 1.0;
    double minFrac = o->min_read_frac;
    int mid = imin((int) (ch->s + (double) window_len * (col + 0.5)),
                   (int) ((ch->s + (double) window_len * col + ch->e) / 2));
    int L = o->mean_read_len;
    int l = imax(mid - L + 1, (int) (-(1 - minFrac) * L));
    int u = imin(mid, (int) (ch->ctg_len - minFrac * L));
    double beta = (double) (u - l) / L;
    if (beta <= 0.25) return 0.25;
    return beta;
}

/* hmm_utils.c:941-947 TruncExponential_getProb */
static double trunc_exp_prob(double lambda, double trunc_point, uint8_t x, double beta) {
    double lam = lambda / beta;
    double b = beta * trunc_point;
    if (x < 0.0 || trunc_point < x) return 0.0;
    return lam * exp(-lam * x) / (1 - exp(-lam * b));
}

/* hmm_utils.c:768-793 Gaussian_getComponentProbs; returns <0 when the reference would exit on NaN */
static int gaussian_comp_probs(const double *mean_, const double *var_, const double *w_, int ncomp,
                               uint8_t x, uint8_t pre_x, double alpha, double beta, double *probs) {
    for (int c = 0; c < ncomp; c++) {
        double mean = (1 - alpha) * mean_[c] + alpha * pre_x;
        mean *= beta;
        double var = var_[c];
        var *= beta;
        double w = w_[c];
        probs[c] = w / (sqrt(var * 2 * OHF_PI)) * exp(-0.5 * pow((x - mean), 2) / var);
        if (probs[c] != probs[c]) return -1;
        if (probs[c] < 1e-40) probs[c] = 1e-40;
    }
    return 0;
}

/* hmm_utils.c:1753-1760 -> 1409-1417 -> 753-758 / 941-947 */
double ohf_emission(const ohf_model *m, int region, int state, uint8_t x, uint8_t pre_x,
                    double alpha, double beta, int *err) {
    const ohf_region *r = &m->regions[region];
    if (state == OHF_STATE_ERR && m->model_type == OHF_MODEL_TRUNC_EXP_GAUSSIAN)
        return trunc_exp_prob(r->lambda, r->trunc_point, x, beta);
    double probs[OHF_MAXCOMP];
    if (m->model_type == OHF_MODEL_NEGATIVE_BINOMIAL) { /* hmm_utils.c:1414-1415 -> 480-485: x only */
        if (ohf_nb_comp_probs(r, state, m->ncomp[state], x, probs) < 0) { if (err) *err = -2; return NAN; }
        double totnb = 0.0;
        for (int c = 0; c < m->ncomp[state]; c++) totnb += probs[c];
        return totnb;
    }
    if (gaussian_comp_probs(r->mean[state], r->var[state], r->weight[state], m->ncomp[state],
                            x, pre_x, alpha, beta, probs) < 0) {
        if (err) *err = -2;
        return NAN;
    }
    double tot = 0.0; /* common.c:188-194 Double_sum1DArray */
    for (int c = 0; c < m->ncomp[state]; c++) tot += probs[c];
    return tot;
}

/* hmm_utils.c:2229-2264: validity of state s (0..4) for one window */
static bool state_valid(const ohf_model *m, int s, uint16_t cov, uint16_t mapq, uint16_t clip) {
    double highMapqRatio = (double) mapq / (0.1 + cov);
    if (s == OHF_STATE_DUP && highMapqRatio > m->max_high_mapq_ratio) return false;
    if (s == OHF_STATE_COL && highMapqRatio < m->min_high_mapq_ratio) return false;
    double highlyClippedRatio = (double) clip / (0.1 + cov);
    if (s == 4 && highlyClippedRatio < m->min_highly_clipped_ratio) return false;
    return true;
}

/* hmm_utils.c:2278-2292 Transition_getProbConditional */
double ohf_trans_cond(const ohf_model *m, int region, int pre, int state,
                      uint16_t cov, uint16_t mapq, uint16_t clip) {
    const ohf_region *r = &m->regions[region];
    double tot = 0.0;
    for (int s = 0; s < OHF_NSTATES + 1; s++)
        if (state_valid(m, s, cov, mapq, clip)) tot += r->trans[pre][s];
    double prob = r->trans[pre][state];
    if (state_valid(m, state, cov, mapq, clip)) return prob / tot;
    return 0.0;
}

static void chunk_alloc(ohf_chunk *ch) {
    if (!ch->f) ch->f = malloc(sizeof(double) * 4 * (size_t) ch->n);
    if (!ch->b) ch->b = malloc(sizeof(double) * 4 * (size_t) ch->n);
    if (!ch->scales) ch->scales = malloc(sizeof(double) * (size_t) ch->n);
}

/* hmm.c:333-434 EM_runForward */
int ohf_chunk_forward(ohf_chunk *ch, const ohf_model *m, int window_len, const ohf_run_opts *o) {
    chunk_alloc(ch);
    int T = ch->n, err = 0;
    double *f = ch->f;
    memset(f, 0, sizeof(double) * 4 * (size_t) T);
    memset(ch->scales, 0, sizeof(double) * (size_t) T);
    ch->loglikelihood = 0.0;
    for (int i = 0; i < T; i++) {
        double beta = ohf_beta(ch, window_len, i, o);
        int region = (uint8_t) region_of(ch->annot[i]);
        uint8_t x = (uint8_t) ch->cov[i];
        double scale = 0.0;
        if (region >= m->n_regions) return -3;
        if (i == 0) { /* hmm.c:333-364 */
            for (int s = 0; s < OHF_NSTATES; s++) {
                double e = ohf_emission(m, region, s, x, 0, 0.0, beta, &err);
                double t = m->regions[region].trans[OHF_NSTATES][s];
                f[s] = e * t;
                scale += f[s];
            }
            ch->scales[0] = scale;
            for (int s = 0; s < OHF_NSTATES; s++) f[s] /= scale;
        } else { /* hmm.c:366-420 */
            int pre_region = (uint8_t) region_of(ch->annot[i - 1]);
            uint8_t pre_x = (uint8_t) ch->cov[i - 1];
            for (int s = 0; s < OHF_NSTATES; s++) {
                for (int p = 0; p < OHF_NSTATES; p++) {
                    double alpha = m->alpha[p][s];
                    double e = ohf_emission(m, region, s, x, pre_x, alpha, beta, &err);
                    double t;
                    if (region != pre_region) t = 1.0 / (OHF_NSTATES + 1);
                    else t = ohf_trans_cond(m, region, p, s, ch->cov[i], ch->mapq[i], ch->clip[i]);
                    f[4 * i + s] += (f[4 * (i - 1) + p] * t * e);
                }
                scale += f[4 * i + s];
            }
            ch->scales[i] = scale;
            if (ch->scales[i] < 1e-50) return -1; /* hmm.c:412-415 */
            for (int s = 0; s < OHF_NSTATES; s++) f[4 * i + s] /= ch->scales[i];
        }
        if (err) return err;
        ch->loglikelihood += log(ch->scales[i]); /* hmm.c:428 */
    }
    return 0;
}

/* hmm.c:452-545 EM_runBackward */
int ohf_chunk_backward(ohf_chunk *ch, const ohf_model *m, int window_len, const ohf_run_opts *o) {
    int T = ch->n, err = 0;
    double *b = ch->b;
    memset(b, 0, sizeof(double) * 4 * (size_t) T);
    for (int i = T - 1; i >= 0; i--) {
        if (i == T - 1) { /* hmm.c:452-467 */
            int region = (uint8_t) region_of(ch->annot[T - 1]);
            for (int s = 0; s < OHF_NSTATES; s++) b[4 * i + s] = m->regions[region].trans[s][OHF_NSTATES];
            for (int s = 0; s < OHF_NSTATES; s++) b[4 * i + s] /= ch->scales[T - 1];
            continue;
        }
        double beta = ohf_beta(ch, window_len, i + 1, o);
        int region = (uint8_t) region_of(ch->annot[i + 1]);
        int pre_region = (uint8_t) region_of(ch->annot[i]);
        uint8_t x = (uint8_t) ch->cov[i + 1];
        uint8_t pre_x = (uint8_t) ch->cov[i];
        for (int s = 0; s < OHF_NSTATES; s++) {
            for (int p = 0; p < OHF_NSTATES; p++) {
                double alpha = m->alpha[p][s];
                double e = ohf_emission(m, region, s, x, pre_x, alpha, beta, &err);
                double t;
                if (region != pre_region) t = 1.0 / (OHF_NSTATES + 1);
                else t = ohf_trans_cond(m, region, p, s, ch->cov[i + 1], ch->mapq[i + 1], ch->clip[i + 1]);
                b[4 * i + p] += t * e * b[4 * (i + 1) + s];
            }
        }
        if (err) return err;
        if (ch->scales[i] < 1e-50) return -1; /* hmm.c:521-524 */
        for (int s = 0; s < OHF_NSTATES; s++) b[4 * i + s] /= ch->scales[i];
    }
    return 0;
}

/* hmm_utils.c:812-839 Gaussian_updateEstimator */
static int gaussian_update(ohf_estimator *est, const double *mean_, const double *var_, const double *w_,
                           int ncomp, uint8_t x, uint8_t pre_x, double alpha, double beta, double count) {
    double x_adjusted = (x - alpha * pre_x) / (1.0 - alpha);
    double probs[OHF_MAXCOMP];
    if (gaussian_comp_probs(mean_, var_, w_, ncomp, x, pre_x, alpha, beta, probs) < 0) return -2;
    double tot = 0.0;
    for (int c = 0; c < ncomp; c++) tot += probs[c];
    for (int c = 0; c < ncomp; c++) {
        double w = count * probs[c] / tot;
        est->num[OHF_P_MEAN][c] += w * x_adjusted; /* hmm_utils.c:58-64 */
        est->den[OHF_P_MEAN][c] += w;
        double z = (x_adjusted - mean_[c]) * (1.0 - alpha);
        est->num[OHF_P_VAR][c] += w * z * z;
        est->den[OHF_P_VAR][c] += w;
        est->num[OHF_P_WEIGHT][c] += w;           /* hmm_utils.c:66-74 */
        for (int i = 0; i < ncomp; i++) est->den[OHF_P_WEIGHT][i] += w;
    }
    return 0;
}

/* hmm.c:563-650 EM_updateEstimators (columns 1..T-2; pair (0,1) and the last column skipped) */
int ohf_chunk_update_estimators(ohf_chunk *ch, const ohf_model *m, ohf_region *acc, int window_len,
                                const ohf_run_opts *o) {
    int T = ch->n, err = 0;
    const double *f = ch->f, *b = ch->b;
    const bool nb = m->model_type == OHF_MODEL_NEGATIVE_BINOMIAL;
    /* count data of the chunk's private model copy: zero at the start of every pass (hmm.c:288-298, hmm_utils.c:1701) */
    double (*counts)[OHF_NSTATES][OHF_MAX_COVERAGE_VALUE] = nb ? calloc((size_t) m->n_regions, sizeof(*counts)) : NULL;
    for (int i = 1; i < T; i++) {
        if (i == T - 1) continue; /* hmm.c:564-566 */
        double beta = ohf_beta(ch, window_len, i + 1, o);
        int region = (uint8_t) region_of(ch->annot[i + 1]);
        int pre_region = (uint8_t) region_of(ch->annot[i]);
        uint8_t x = (uint8_t) ch->cov[i + 1];
        uint8_t pre_x = (uint8_t) ch->cov[i];
        const ohf_region *r = &m->regions[region];
        ohf_region *a = &acc[region];
        for (int s = 0; s < OHF_NSTATES; s++) {
            for (int p = 0; p < OHF_NSTATES; p++) {
                double alpha = m->alpha[p][s];
                double e = ohf_emission(m, region, s, x, pre_x, alpha, beta, &err);
                double t;
                if (region != pre_region) t = 1.0 / (OHF_NSTATES + 1);
                else t = ohf_trans_cond(m, region, p, s, ch->cov[i + 1], ch->mapq[i + 1], ch->clip[i + 1]);
                double count = f[4 * i + p] * t * e * b[4 * (i + 1) + s];
                double adjusted = count / OHF_TERMINATION_PROB; /* hmm.c:614 */
                if (nb) { /* hmm.c:615-617, count_data.c:49-57 */
                    counts[region][s][x < OHF_MAX_COVERAGE_VALUE ? x : OHF_MAX_COVERAGE_VALUE - 1] += adjusted;
                } else if (s == OHF_STATE_ERR && m->model_type == OHF_MODEL_TRUNC_EXP_GAUSSIAN) {
                    a->est[s].num[0][0] += adjusted * x; /* hmm_utils.c:1027-1034 */
                    a->est[s].den[0][0] += adjusted;
                } else {
                    if (gaussian_update(&a->est[s], r->mean[s], r->var[s], r->weight[s], m->ncomp[s],
                                        x, pre_x, alpha, beta, adjusted) < 0) err = -2;
                }
                a->count[p][s] += adjusted; /* hmm_utils.c:2010-2015 */
            }
        }
        if (err) { free(counts); return err; }
    }
    if (nb) { /* hmm.c:644-649 */
        for (int region = 0; region < m->n_regions && !err; region++)
            if (ohf_nb_update_from_counts(&acc[region], &m->regions[region], m->ncomp, counts[region]) < 0) err = -2;
        free(counts);
    }
    return err;
}

/* hmm.c:671-685 EM_getPosterior */
void ohf_posterior(const ohf_chunk *ch, int pos, double post[4]) {
    double total = 0.0;
    for (int s = 0; s < OHF_NSTATES; s++) {
        post[s] = ch->f[4 * pos + s] * ch->b[4 * pos + s] * ch->scales[pos];
        total += post[s];
    }
    for (int s = 0; s < OHF_NSTATES; s++) post[s] /= total;
}

/* hmm.c:687-692 + common.c:292-304 (first maximum wins) */
int ohf_most_probable_state(const ohf_chunk *ch, int pos) {
    double post[4];
    ohf_posterior(ch, pos, post);
    double maxv = post[0];
    int index = 0;
    for (int i = 0; i < OHF_NSTATES; i++)
        if (maxv < post[i]) { maxv = post[i]; index = i; }
    return index;
}

/* ---- EM_runOneIterationForList, hmm.c:739-763 (thread pool over chunks, then a sequential
 *      in-list-order reduction of loglikelihood and estimators) ---- */
typedef struct {
    ohf_chunks *cc; const ohf_model *m; const ohf_run_opts *o; int forward_only;
    ohf_region *per_chunk_acc; /* [n_chunks][n_regions] */
    int next; int status; pthread_mutex_t mu;
} work_t;

static void *worker(void *arg_) {
    work_t *w = arg_;
    for (;;) {
        pthread_mutex_lock(&w->mu);
        int c = w->next++;
        pthread_mutex_unlock(&w->mu);
        if (c >= w->cc->n_chunks) break;
        ohf_chunk *ch = &w->cc->chunks[c];
        int st = ohf_chunk_forward(ch, w->m, w->cc->window_len, w->o); /* hmm.c:719 */
        if (st == 0 && !w->forward_only) {
            st = ohf_chunk_backward(ch, w->m, w->cc->window_len, w->o); /* hmm.c:720 */
            if (st == 0)
                st = ohf_chunk_update_estimators(ch, w->m, w->per_chunk_acc + (size_t) c * w->m->n_regions,
                                                 w->cc->window_len, w->o); /* hmm.c:721 */
            if (st == 0)
                for (int pos = 0; pos < ch->n; pos++) /* hmm.c:730-736 */
                    ch->prediction[pos] = (int8_t) ohf_most_probable_state(ch, pos);
        }
        if (st != 0) { pthread_mutex_lock(&w->mu); if (!w->status) w->status = st; pthread_mutex_unlock(&w->mu); }
    }
    return NULL;
}

int ohf_run_iteration(ohf_chunks *cc, ohf_model *m, const ohf_run_opts *o, int forward_only) {
    int R = m->n_regions, C = cc->n_chunks;
    work_t w = { cc, m, o, forward_only, NULL, 0, 0, PTHREAD_MUTEX_INITIALIZER };
    w.per_chunk_acc = calloc((size_t) C * R, sizeof(ohf_region)); /* zeroed private copies, hmm.c:288-298 */
    int nt = o->threads < 1 ? 1 : o->threads;
    pthread_t *th = malloc(sizeof(pthread_t) * nt);
    for (int i = 0; i < nt; i++) pthread_create(&th[i], NULL, worker, &w);
    for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
    free(th);
    m->loglikelihood = 0.0;
    for (int c = 0; c < C; c++) { /* hmm.c:759-763, 548-560; hmm_utils.c:49-56, 2000-2008 */
        m->loglikelihood += cc->chunks[c].loglikelihood;
        if (forward_only) continue;
        for (int r = 0; r < R; r++) {
            ohf_region *dst = &m->regions[r];
            const ohf_region *src = &w.per_chunk_acc[(size_t) c * R + r];
            for (int s = 0; s < OHF_NSTATES; s++) {
                int np = (s == OHF_STATE_ERR && m->model_type == OHF_MODEL_TRUNC_EXP_GAUSSIAN) ? 1 : 3;
                int nc = (np == 1) ? 1 : m->ncomp[s];
                for (int p = 0; p < np; p++)
                    for (int k = 0; k < nc; k++) {
                        dst->est[s].num[p][k] += src->est[s].num[p][k];
                        dst->est[s].den[p][k] += src->est[s].den[p][k];
                    }
            }
            for (int s1 = 0; s1 < OHF_NSTATES; s1++)
                for (int s2 = 0; s2 < OHF_NSTATES; s2++) dst->count[s1][s2] += src->count[s1][s2];
        }
    }
    free(w.per_chunk_acc);
    return w.status;
}
