#include <stdio.h>
/*
 * ohf_squarem.c — ORACLE (test infrastructure, not product code).
 * SQUAREM acceleration restated from /root/reference/programs/submodules/hmm/hmm.c:820-1098 and
 * the accelerated branch of runHMMFlagger (/root/reference/programs/src/hmm_flagger.c:382-416).
 */
#include "ohf.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static ohf_model *model_copy(const ohf_model *src) { /* hmm.c:96-110 */
    ohf_model *m = malloc(sizeof(ohf_model));
    *m = *src;
    m->regions = malloc(sizeof(ohf_region) * (size_t) src->n_regions);
    memcpy(m->regions, src->regions, sizeof(ohf_region) * (size_t) src->n_regions);
    return m;
}

static void model_assign(ohf_model *dst, const ohf_model *src) {
    ohf_region *keep = dst->regions;
    *dst = *src;
    dst->regions = keep;
    memcpy(dst->regions, src->regions, sizeof(ohf_region) * (size_t) src->n_regions);
}

/* parameter slots in the reference's iterator order (hmm_utils.c:1111-1249): per region, per state, per
 * component: mean, var, weight (Gaussian) | lambda (trunc-exp; the trunc point is not iterated); then the
 * 4x4 transition block (hmm.c:976-992) */
static int collect(ohf_model *m, double **slots) {
    int n = 0;
    for (int r = 0; r < m->n_regions; r++) {
        ohf_region *g = &m->regions[r];
        for (int s = 0; s < OHF_NSTATES; s++) {
            if (s == OHF_STATE_ERR && m->model_type == OHF_MODEL_TRUNC_EXP_GAUSSIAN) { slots[n++] = &g->lambda; continue; }
            if (m->model_type == OHF_MODEL_NEGATIVE_BINOMIAL) { /* theta, lambda, weight (hmm_utils.c:1122-1131) */
                for (int c = 0; c < m->ncomp[s]; c++) { slots[n++] = &g->theta[s][c]; slots[n++] = &g->nb_lambda[s][c]; slots[n++] = &g->weight[s][c]; }
                continue;
            }
            for (int c = 0; c < m->ncomp[s]; c++) { slots[n++] = &g->mean[s][c]; slots[n++] = &g->var[s][c]; slots[n++] = &g->weight[s][c]; }
        }
        for (int i = 0; i < OHF_NSTATES; i++) for (int j = 0; j < OHF_NSTATES; j++) slots[n++] = &g->trans[i][j];
    }
    return n;
}

static bool feasible(const ohf_model *m) { /* hmm.c:80-87; hmm_utils.c:685-694, 920-925, 2130-2139 */
    bool ok = true;
    for (int r = 0; r < m->n_regions; r++) {
        const ohf_region *g = &m->regions[r];
        for (int s = 0; s < OHF_NSTATES; s++) {
            if (s == OHF_STATE_ERR && m->model_type == OHF_MODEL_TRUNC_EXP_GAUSSIAN) { ok &= 0 < g->lambda; ok &= 0 < g->trunc_point; continue; }
            if (m->model_type == OHF_MODEL_NEGATIVE_BINOMIAL) { /* hmm_utils.c:367-376 */
                for (int c = 0; c < m->ncomp[s]; c++) {
                    ok &= (0 < g->theta[s][c]) && (g->theta[s][c] < 1);
                    ok &= (0 < g->nb_lambda[s][c]);
                    ok &= (0 <= g->weight[s][c]) && (g->weight[s][c] <= 1);
                }
                continue;
            }
            for (int c = 0; c < m->ncomp[s]; c++) {
                ok &= (0 < g->mean[s][c]); ok &= (0 < g->var[s][c]);
                ok &= (0 <= g->weight[s][c]) && (g->weight[s][c] <= 1);
            }
        }
        for (int i = 0; i < OHF_NSTATES; i++)
            for (int j = 0; j < OHF_NSTATES; j++)
                if (g->trans[i][j] < 0 || 1 < g->trans[i][j]) return false;
    }
    return ok;
}

typedef struct { ohf_model *m0, *prime, *rr, *rv; double alpha; } accel_t;

static void compute_values(accel_t *a) { /* hmm.c:921-997 */
    enum { MAXS = OHF_MAXREGIONS * (1 + 3 * 3 * OHF_MAXCOMP + 16) };
    static double *s0[MAXS], *sp[MAXS], *sr[MAXS], *sv[MAXS];
    int n = collect(a->m0, s0); collect(a->prime, sp); collect(a->rr, sr); collect(a->rv, sv);
    for (int i = 0; i < n; i++) *sp[i] = *s0[i] - 2 * *sr[i] * a->alpha + *sv[i] * pow(a->alpha, 2);
    ohf_model *m = a->prime;
    for (int r = 0; r < m->n_regions; r++) { /* hmm.c:89-94 */
        ohf_region *g = &m->regions[r];
        for (int s = 0; s < OHF_NSTATES; s++) {
            if (s == OHF_STATE_ERR && m->model_type == OHF_MODEL_TRUNC_EXP_GAUSSIAN) continue;
            double sum = 0.0; /* hmm_utils.c:675-683 */
            for (int c = 0; c < m->ncomp[s]; c++) sum += g->weight[s][c];
            if (0.0 < sum) { double k = 1.0 / sum; for (int c = 0; c < m->ncomp[s]; c++) g->weight[s][c] *= k; }
        }
        for (int i = 0; i < OHF_NSTATES; i++) { /* hmm_utils.c:2165-2183 */
            double row = 0.0;
            for (int j = 0; j < OHF_NSTATES; j++) row += g->trans[i][j];
            for (int j = 0; j < OHF_NSTATES; j++) g->trans[i][j] = g->trans[i][j] / row * (1.0 - OHF_TERMINATION_PROB);
        }
        for (int i = 0; i < OHF_NSTATES; i++) g->trans[i][OHF_NSTATES] = OHF_TERMINATION_PROB;
        g->trans[OHF_NSTATES][OHF_NSTATES] = 0.0;
    }
}

static void shrink(accel_t *a, double margin) { /* hmm.c:871-884 */
    a->alpha = (a->alpha - 1) / 2;
    if (a->alpha > (-1 - margin)) { a->alpha = -1.0; model_assign(a->prime, a->m0); return; }
    compute_values(a);
}

/* the accelerated branch, hmm_flagger.c:382-416: on entry the model holds the statistics of its E-step;
 * on exit it is model prime with the statistics of model prime's E-step */
int ohf_squarem_iteration(ohf_chunks *cc, ohf_model *m, const ohf_run_opts *o, double tol, double *alpha_out) {
    enum { MAXS = OHF_MAXREGIONS * (1 + 3 * 3 * OHF_MAXCOMP + 16) };
    static double *s0[MAXS], *s1[MAXS], *s2[MAXS], *sr[MAXS], *sv[MAXS];
    accel_t a;
    a.m0 = model_copy(m); a.rr = model_copy(m); a.rv = model_copy(m); a.prime = model_copy(m); /* hmm.c:855-860 */
    ohf_estimate_parameters(m, tol);
    ohf_model *m1 = model_copy(m);
    ohf_reset_estimators(m);
    int st = ohf_run_iteration(cc, m, o, 0);
    if (st) return st;
    ohf_estimate_parameters(m, tol);
    ohf_model *m2 = m;
    int n = collect(a.m0, s0); collect(m1, s1); collect(m2, s2); collect(a.rr, sr); collect(a.rv, sv);
    double num = 0.0, den = 0.0; /* hmm.c:999-1098 */
    for (int i = 0; i < n; i++) {
        double r = *s1[i] - *s0[i];
        double v = *s2[i] - *s1[i] - r;
        num += pow(r, 2); den += pow(v, 2);
        *sr[i] = r; *sv[i] = v;
    }
    a.alpha = -1 * sqrt(num / den);
    if (a.alpha > -1) a.alpha = -1;
    double ll0 = a.m0->loglikelihood; /* hmm.c:886-918 */
    compute_values(&a);
    while (!feasible(a.prime)) shrink(&a, 1e-2);
    st = ohf_run_iteration(cc, a.prime, o, 1);
    while (!st && a.prime->loglikelihood < ll0) {
        shrink(&a, 1e-2);
        while (!feasible(a.prime)) shrink(&a, 1e-2);
        st = ohf_run_iteration(cc, a.prime, o, 1);
    }
    if (!st) {
        ohf_reset_estimators(a.prime);
        st = ohf_run_iteration(cc, a.prime, o, 0);
        model_assign(m, a.prime);
    }
    if (alpha_out) *alpha_out = a.alpha;
    if (!st) fprintf(stderr, "Computed alpha rate for accelerating EM = %.4f\n", a.alpha);   /* hmm.c:916 */
    ohf_model_destroy(a.m0); ohf_model_destroy(a.rr); ohf_model_destroy(a.rv); ohf_model_destroy(a.prime); ohf_model_destroy(m1);
    return st;
}
