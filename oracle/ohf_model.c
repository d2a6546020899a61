/*
 * ohf_model.c — ORACLE (test infrastructure, not product code).
 * Initial model and M-step of HMM-Flagger, restated from the reference
 * (citations: file:line under /root/reference/programs/).
 */
#include "ohf.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MIN_COUNT_FOR_PARAMETER_UPDATE 10 /* submodules/hmm_utils/hmm_utils.h:11 */
#define EXP_TRUNC_POINT_COV_FRACTION 0.25 /* hmm_utils.h:12 */
#define ERR_COMP_BINDING_COEF 0.1         /* hmm_utils.h:14 */
#define TRANSITION_INITIAL_DIAG_PROB 0.99 /* submodules/hmm/hmm.c:15 */
#define TRANSITION_PSEUDO_COUNT_VALUE 0.001 /* hmm.c:16 */

/* src/hmm_flagger.c:164-237 createModel + hmm.c:22-77 HMM_construct +
 * hmm_utils.c:1605-1652 EmissionDistSeries_constructForModel + hmm_utils.c:2109-2128 */
ohf_model *ohf_model_create(int model_type, int n_collapsed, const int32_t *region_coverages,
                            int n_regions, bool start_only, int avg_alignment_len, int window_len,
                            const double alpha[4][4], double max_high_mapq_ratio,
                            double min_high_mapq_ratio) {
    if (n_collapsed < 1 || n_collapsed > OHF_MAXCOMP || n_regions < 1 || n_regions > OHF_MAXREGIONS) return NULL;
    ohf_model *m = calloc(1, sizeof(ohf_model));
    m->model_type = model_type;
    m->n_regions = n_regions;
    m->ncomp[OHF_STATE_ERR] = 1;
    m->ncomp[OHF_STATE_DUP] = 1;
    m->ncomp[OHF_STATE_HAP] = 1;
    m->ncomp[OHF_STATE_COL] = n_collapsed;
    memcpy(m->alpha, alpha, sizeof(m->alpha));
    m->max_high_mapq_ratio = max_high_mapq_ratio;
    m->min_high_mapq_ratio = min_high_mapq_ratio;
    m->min_highly_clipped_ratio = 1.0; /* hmm_flagger.c:222 */
    m->regions = calloc((size_t) n_regions, sizeof(ohf_region));

    double medianCoverage = region_coverages[0]; /* hmm_flagger.c:189-195 */
    if (start_only) medianCoverage *= (double) window_len / avg_alignment_len;
    /* hmm_flagger.c:213-220 with initialRandomDev = 0 (getRandomNumber(1,1) == 1.0, :113-116) */
    double means[OHF_NSTATES][OHF_MAXCOMP];
    memset(means, 0, sizeof(means));
    means[OHF_STATE_ERR][0] = medianCoverage * ERR_COMP_BINDING_COEF * 1.0;
    means[OHF_STATE_DUP][0] = medianCoverage * 0.5 * 1.0;
    means[OHF_STATE_HAP][0] = medianCoverage * 1.0 * 1.0;
    for (int i = 0; i < n_collapsed; i++) means[OHF_STATE_COL][i] = means[OHF_STATE_HAP][0] * (i + 2) * 1.0;

    for (int r = 0; r < n_regions; r++) {
        ohf_region *g = &m->regions[r];
        double scale = (double) region_coverages[r] / medianCoverage; /* hmm_flagger.c:199 */
        double mr[OHF_NSTATES][OHF_MAXCOMP];
        for (int s = 0; s < OHF_NSTATES; s++)
            for (int c = 0; c < OHF_MAXCOMP; c++) mr[s][c] = means[s][c] * scale; /* hmm.c:43-47 */
        g->lambda = 1.0; /* hmm_utils.c:1620 */
        g->trunc_point = mr[OHF_STATE_HAP][0] * EXP_TRUNC_POINT_COV_FRACTION;
        for (int s = 0; s < OHF_NSTATES; s++)
            for (int c = 0; c < m->ncomp[s]; c++) { /* hmm_utils.c:733-741, 658-673 */
                g->mean[s][c] = mr[s][c];
                g->var[s][c] = mr[s][c] * 1.0;
                g->weight[s][c] = 1.0 / m->ncomp[s];
            }
        if (model_type == OHF_MODEL_NEGATIVE_BINOMIAL) /* hmm_utils.c:1635-1639 */
            for (int s = 0; s < OHF_NSTATES; s++) ohf_nb_init(g, s, mr[s], m->ncomp[s]);
        /* hmm_utils.c:2109-2128 Transition_constructSymmetricBiased(4, 0.99) */
        double term = OHF_TERMINATION_PROB;
        for (int i = 0; i < 5; i++)
            for (int j = 0; j < 5; j++) {
                g->trans[i][j] = (1.0 - TRANSITION_INITIAL_DIAG_PROB) / (OHF_NSTATES - 1) * (1.0 - term);
                g->pseudo[i][j] = TRANSITION_PSEUDO_COUNT_VALUE; /* hmm.c:59-60 */
            }
        for (int i = 0; i < 5; i++) g->trans[i][i] = TRANSITION_INITIAL_DIAG_PROB * (1.0 - term);
        for (int s = 0; s < OHF_NSTATES; s++) {
            g->trans[OHF_NSTATES][s] = 1.0 / OHF_NSTATES;
            g->trans[s][OHF_NSTATES] = term;
        }
        g->trans[OHF_NSTATES][OHF_NSTATES] = 0.0;
    }
    return m;
}

void ohf_model_destroy(ohf_model *m) {
    if (!m) return;
    free(m->regions);
    free(m);
}

/* hmm.c:129-134 */
void ohf_reset_estimators(ohf_model *m) {
    for (int r = 0; r < m->n_regions; r++) {
        memset(m->regions[r].est, 0, sizeof(m->regions[r].est));
        memset(m->regions[r].count, 0, sizeof(m->regions[r].count));
    }
}

/* hmm_flagger.c:105-111 + 1012-1013 */
int ohf_best_collapsed_comps(const ohf_chunks *cc) {
    int maxCoverage = 0;
    for (int c = 0; c < cc->n_chunks; c++)
        for (int i = 0; i < cc->chunks[c].n; i++)
            if (maxCoverage < cc->chunks[c].cov[i]) maxCoverage = cc->chunks[c].cov[i];
    int minRegion = cc->region_coverages[0];
    for (int r = 1; r < cc->n_regions; r++)
        if (cc->region_coverages[r] < minRegion) minRegion = cc->region_coverages[r];
    if (minRegion == 0) return -1;
    int k = maxCoverage / minRegion + 1;
    k = k < 2 ? 2 : k;
    k = k > 10 ? 10 : k;
    return k;
}

/* hmm_utils.c:949-956 */
static double trunc_exp_ll(double lam, double b, double num, double den) {
    return den * log(lam) - den * log(1.0 - exp(-lam * b)) - num * lam;
}

/* hmm_utils.c:969-1011 TruncExponential_estimateLambda (golden section on [0, truncPoint]) */
double ohf_estimate_lambda(double trunc_point, double num, double den, double tol) {
    double a = 0.0, b = trunc_point;
    double invphi = (sqrt(5.0) - 1.0) / 2.0;
    double invphi2 = (3.0 - sqrt(5.0)) / 2.0;
    double h = b - a;
    if (h <= tol) return (b + a) / 2.0;
    int n = ceil(log(tol / h) / log(invphi));
    double c = a + invphi2 * h;
    double d = a + invphi * h;
    double yc = trunc_exp_ll(c, trunc_point, num, den);
    double yd = trunc_exp_ll(d, trunc_point, num, den);
    for (int k = 0; k < n - 1; k++) {
        if (yc > yd) {
            b = d; d = c; yd = yc;
            h = invphi * h;
            c = a + invphi2 * h;
            yc = trunc_exp_ll(c, trunc_point, num, den);
        } else {
            a = c; c = d; yc = yd;
            h = invphi * h;
            d = a + invphi * h;
            yd = trunc_exp_ll(d, trunc_point, num, den);
        }
    }
    if (yc > yd) return (a + d) / 2.0;
    return (c + b) / 2.0;
}

/* binding coefficient of (state, param, comp): hmm_utils.c:191-238 (Gaussian), 290-304 */
static double binding_coef(const ohf_model *m, int s, int p, int c) {
    if (p == OHF_P_WEIGHT) return 0.0;
    switch (s) {
        case OHF_STATE_ERR: return ERR_COMP_BINDING_COEF;
        case OHF_STATE_DUP: return 0.5;
        case OHF_STATE_HAP: return 1.0;
        default: return 2.0 + (double) c * 1.0; /* hmm_utils.c:143-157 */
    }
    (void) m;
}

static bool is_gaussian_state(const ohf_model *m, int s) {
    return !(s == OHF_STATE_ERR && m->model_type == OHF_MODEL_TRUNC_EXP_GAUSSIAN);
}

/* hmm_utils.c:842-859 Gaussian_updateParameter */
static bool gaussian_update_param(ohf_region *g, int s, int p, int c, double value, double tol) {
    double *slot = p == OHF_P_MEAN ? &g->mean[s][c] : p == OHF_P_VAR ? &g->var[s][c] : &g->weight[s][c];
    double oldValue = *slot;
    *slot = value;
    double diffRatio = 1.0e-4 < oldValue ? fabs(value / oldValue - 1.0) : 0.0;
    return diffRatio < tol;
}

/* hmm_utils.c:1817-1858 EmissionDistSeries_estimateOneParameterType for DIST_GAUSSIAN,
 * with 1791-1815 (bound estimator) and 76-92 (getEstimation) */
static bool estimate_gaussian_param(const ohf_model *m, ohf_region *g, int p, double tol) {
    bool converged = true;
    double bnum = 0.0, bden = 0.0;
    for (int s = 0; s < OHF_NSTATES; s++) {
        if (!is_gaussian_state(m, s)) continue;
        for (int c = 0; c < m->ncomp[s]; c++) {
            double factor = binding_coef(m, s, p, c);
            if (0.0 < factor) {
                bnum += g->est[s].num[p][c] / factor;
                bden += g->est[s].den[p][c];
            }
        }
    }
    double boundCount = bden;
    double boundEstimation = (bden == 0) ? 0.0 : bnum / bden;
    for (int s = 0; s < OHF_NSTATES; s++) {
        if (!is_gaussian_state(m, s)) continue;
        for (int c = 0; c < m->ncomp[s]; c++) {
            double factor = binding_coef(m, s, p, c);
            double estimation, count;
            if (0.0 < factor) {
                estimation = boundEstimation * factor;
                count = boundCount;
            } else {
                count = g->est[s].den[p][c];
                estimation = (count == 0) ? 0.0 : g->est[s].num[p][c] / g->est[s].den[p][c];
            }
            if (MIN_COUNT_FOR_PARAMETER_UPDATE < count)
                converged &= gaussian_update_param(g, s, p, c, estimation, tol);
        }
    }
    return converged;
}

/* hmm_utils.c:1860-1903 EmissionDistSeries_estimateParameters */
static bool estimate_emissions(const ohf_model *m, ohf_region *g, double tol) {
    bool converged = true;
    if (m->model_type == OHF_MODEL_NEGATIVE_BINOMIAL) return ohf_nb_estimate(m, g, tol); /* hmm_utils.c:1885-1900 */
    for (int p = 0; p < 3; p++) converged &= estimate_gaussian_param(m, g, p, tol);
    if (m->model_type == OHF_MODEL_TRUNC_EXP_GAUSSIAN) {
        /* binding coefficient 0 => own estimator; golden-section with the OLD trunc point */
        double num = g->est[OHF_STATE_ERR].num[0][0], den = g->est[OHF_STATE_ERR].den[0][0];
        double count = den;
        double estimation = (den == 0) ? 0.0 : ohf_estimate_lambda(g->trunc_point, num, den, 1e-6);
        if (MIN_COUNT_FOR_PARAMETER_UPDATE < count) { /* hmm_utils.c:1036-1054 */
            double oldLambda = g->lambda;
            g->lambda = estimation;
            double diffRatio = 1.0e-4 < oldLambda ? fabs(estimation / oldLambda - 1.0) : 0.0;
            converged &= diffRatio < tol;
        }
        g->trunc_point = g->mean[OHF_STATE_HAP][0] * EXP_TRUNC_POINT_COV_FRACTION; /* :1878-1882 */
    }
    return converged;
}

/* hmm_utils.c:2185-2219 Transition_estimateTransitionMatrix */
static bool estimate_transitions(ohf_region *g, double tol) {
    bool converged = true;
    double term = OHF_TERMINATION_PROB;
    for (int i1 = 0; i1 < OHF_NSTATES; i1++) {
        double rowSum = 0.0;
        for (int i2 = 0; i2 < OHF_NSTATES; i2++) rowSum += g->count[i1][i2] + g->pseudo[i1][i2];
        for (int i2 = 0; i2 < OHF_NSTATES; i2++) {
            double oldValue = g->trans[i1][i2];
            double newValue = (g->count[i1][i2] + g->pseudo[i1][i2]) / rowSum * (1.0 - term);
            g->trans[i1][i2] = newValue;
            double diffRatio = 1.0e-6 < oldValue ? fabs(newValue / oldValue - 1.0) : 0.0;
            converged &= diffRatio < tol;
        }
    }
    for (int i1 = 0; i1 < OHF_NSTATES; i1++) g->trans[i1][OHF_NSTATES] = term;
    for (int i2 = 0; i2 < OHF_NSTATES; i2++) g->trans[OHF_NSTATES][i2] = 1.0 / OHF_NSTATES;
    g->trans[OHF_NSTATES][OHF_NSTATES] = 0.0;
    return converged;
}

/* hmm.c:120-127 HMM_estimateParameters */
bool ohf_estimate_parameters(ohf_model *m, double tol) {
    bool converged = true;
    for (int r = 0; r < m->n_regions; r++) {
        converged &= estimate_emissions(m, &m->regions[r], tol);
        converged &= estimate_transitions(&m->regions[r], tol);
    }
    return converged;
}
