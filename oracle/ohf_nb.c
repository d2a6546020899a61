/*
 * ohf_nb.c — ORACLE (test infrastructure, not product code).  PARITY UNPINNED (see ohf.h).
 * The negative_binomial model type, restated from the reference
 * (citations: file:line under /root/reference/programs/submodules/).
 */
#include "ohf.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* digamma/digamma.c:27-117 — digamma in long double: reflection below 0, recurrence below 1, exact values at 1, 2, 3,
 * duplication formula above 3, and on (1,3) the Chebyshev expansion of R. J. Mathar, arXiv:math.CA/0403344 app. E
 * (J. Wimp, Math. Comp. 15 (1961) 174, table 1).  The coefficients are the published table. */
long double ohf_digammal(long double x) {
    static const long double euler = 0.5772156649015328606065120900824024L;
    static const long double ln2 = 0.6931471805599453094172321214581766L;
    static const long double pi = 3.1415926535897932384626433832795029L;
    static const long double K[] = {
        .30459198558715155634315638246624251L, .72037977439182833573548891941219706L,
        -.12454959243861367729528855995001087L, .27769457331927827002810119567456810e-1L,
        -.67762371439822456447373550186163070e-2L, .17238755142247705209823876688592170e-2L,
        -.44817699064252933515310345718960928e-3L, .11793660000155572716272710617753373e-3L,
        -.31253894280980134452125172274246963e-4L, .83173997012173283398932708991137488e-5L,
        -.22191427643780045431149221890172210e-5L, .59302266729329346291029599913617915e-6L,
        -.15863051191470655433559920279603632e-6L, .42459203983193603241777510648681429e-7L,
        -.11369129616951114238848106591780146e-7L, .304502217295931698401459168423403510e-8L,
        -.81568455080753152802915013641723686e-9L, .21852324749975455125936715817306383e-9L,
        -.58546491441689515680751900276454407e-10L, .15686348450871204869813586459513648e-10L,
        -.42029496273143231373796179302482033e-11L, .11261435719264907097227520956710754e-11L,
        -.30174353636860279765375177200637590e-12L, .80850955256389526647406571868193768e-13L,
        -.21663779809421233144009565199997351e-13L, .58047634271339391495076374966835526e-14L,
        -.15553767189204733561108869588173845e-14L, .41676108598040807753707828039353330e-15L,
        -.11167065064221317094734023242188463e-15L };
    if (x < 0.0L) return ohf_digammal(1.0L - x) + pi / tanl(pi * (1.0L - x));
    if (x < 1.0L) return ohf_digammal(1.0L + x) - 1.0L / x;
    if (x == 1.0L) return -euler;
    if (x == 2.0L) return 1.0L - euler;
    if (x == 3.0L) return 1.5L - euler;
    if (x > 3.0L) return 0.5L * (ohf_digammal(x / 2.0L) + ohf_digammal((x + 1.0L) / 2.0L)) + ln2;
    long double t0 = 1.0L, t1 = x - 2.0L;              /* T_0, T_1 of the shifted argument */
    long double res = K[0] + K[1] * t1;
    x -= 2.0L;
    for (int n = 2; n < (int) (sizeof(K) / sizeof(K[0])); n++) {
        const long double t2 = 2.0L * x * t1 - t0;     /* Chebyshev recursion */
        res += K[n] * t2;
        t0 = t1;
        t1 = t2;
    }
    return res;
}

double ohf_nb_r(double theta, double lambda) { return -1 * lambda / log(theta); }            /* hmm_utils.c:458-461 */
double ohf_nb_mean(double theta, double lambda) {                                             /* :463-467 */
    double r = -1 * lambda / log(theta);
    return r * (1 - theta) / theta;
}
double ohf_nb_var(double theta, double lambda) {                                              /* :469-473 */
    double r = -1 * lambda / log(theta);
    return r * (1 - theta) / pow(theta, 2);
}

/* hmm_utils.c:428-456, 446-456: NegativeBinomial_constructByMean(mean, 1.5, n) */
void ohf_nb_init(ohf_region *g, int s, const double *mean, int ncomp) {
    for (int c = 0; c < ncomp; c++) {
        double var = mean[c] * 1.5;
        g->theta[s][c] = mean[c] / var;                                                       /* :448-451 */
        double r = pow(mean[c], 2) / (var - mean[c]);                                         /* :453-457 */
        g->nb_lambda[s][c] = -1 * r * log(mean[c] / var);
        g->weight[s][c] = 1.0 / ncomp;
    }
}

/* hmm_utils.c:497-520 NegativeBinomial_getComponentProbs; returns <0 where the reference exits */
int ohf_nb_comp_probs(const ohf_region *g, int s, int ncomp, uint8_t x, double *probs) {
    for (int c = 0; c < ncomp; c++) {
        double theta = g->theta[s][c];
        double r = ohf_nb_r(g->theta[s][c], g->nb_lambda[s][c]);
        double w = g->weight[s][c];
        probs[c] = w * exp(lgamma(r + x) - lgamma(r) - lgamma(x + 1) + r * log(theta) + (double) x * log(1 - theta));
        if (probs[c] != probs[c]) return -2;
        if (probs[c] < 1e-40) probs[c] = 1e-40;
    }
    return 0;
}

/* hmm_utils.c:394-408 NegativeBinomial_fillDigammaTable: table[comp][0..250] */
void ohf_nb_digamma_table(const ohf_region *g, int s, int ncomp, double table[][OHF_MAX_COVERAGE_VALUE + 1]) {
    for (int c = 0; c < ncomp; c++) {
        double r = ohf_nb_r(g->theta[s][c], g->nb_lambda[s][c]);
        table[c][0] = (double) ohf_digammal(r);
        for (int x = 1; x <= OHF_MAX_COVERAGE_VALUE; x++) table[c][x] = table[c][x - 1] + 1.0 / (r + x - 1);
    }
}

/* hmm_utils.c:537-566 NegativeBinomial_updateEstimator */
int ohf_nb_update(ohf_estimator *est, const ohf_region *g, int s, int ncomp,
                  double table[][OHF_MAX_COVERAGE_VALUE + 1], uint8_t x, double count) {
    double probs[OHF_MAXCOMP];
    if (ohf_nb_comp_probs(g, s, ncomp, x, probs) < 0) return -2;
    double tot = 0.0;
    for (int c = 0; c < ncomp; c++) tot += probs[c];
    for (int c = 0; c < ncomp; c++) {
        double theta = g->theta[s][c];
        double r = ohf_nb_r(g->theta[s][c], g->nb_lambda[s][c]);
        double beta = -1 * theta / (1 - theta) - 1 / log(theta);
        double w = count * probs[c] / tot;
        double delta = r * (table[c][x] - table[c][0]);
        est->num[OHF_P_NB_LAMBDA][c] += w * delta;
        est->den[OHF_P_NB_LAMBDA][c] += w;
        est->num[OHF_P_NB_THETA][c] += w * delta * beta;
        est->den[OHF_P_NB_THETA][c] += w * delta * beta + w * (x - delta);
        est->num[OHF_P_WEIGHT][c] += w;                                                       /* :66-74 */
        for (int i = 0; i < ncomp; i++) est->den[OHF_P_WEIGHT][i] += w;
    }
    return 0;
}

/* hmm_utils.c:1661-1673 EmissionDistSeries_updateAllEstimatorsUsingCountData for one region:
 * counts[state][0..249]; x = 250 was folded into the last bin by CountData_increment (count_data.c:49-57) */
int ohf_nb_update_from_counts(ohf_region *acc, const ohf_region *g, const int *ncomp,
                              double counts[OHF_NSTATES][OHF_MAX_COVERAGE_VALUE]) {
    static _Thread_local double table[OHF_MAXCOMP][OHF_MAX_COVERAGE_VALUE + 1];
    for (int s = 0; s < OHF_NSTATES; s++) {
        ohf_nb_digamma_table(g, s, ncomp[s], table);
        for (int x = 0; x < OHF_MAX_COVERAGE_VALUE; x++) {
            double count = counts[s][x];
            if (0 < count)
                if (ohf_nb_update(&acc->est[s], g, s, ncomp[s], table, (uint8_t) x, count) < 0) return -2;
        }
    }
    return 0;
}

/* binding coefficients, hmm_utils.c:240-288 ParameterBinding_getDefault1DArrayForNegativeBinomial */
static double nb_binding_coef(int s, int p, int c) {
    if (p == OHF_P_WEIGHT) return 0.0;
    if (p == OHF_P_NB_THETA) return 1.0;
    switch (s) {
        case OHF_STATE_ERR: return 0.1;          /* ERR_COMP_BINDING_COEF, hmm_utils.h:14 */
        case OHF_STATE_DUP: return 0.5;
        case OHF_STATE_HAP: return 1.0;
        default: return 2.0 + (double) c * 1.0;
    }
}

/* hmm_utils.c:568-587 NegativeBinomial_updateParameter */
static bool nb_update_param(ohf_region *g, int s, int p, int c, double value, double tol) {
    double *slot = p == OHF_P_NB_THETA ? &g->theta[s][c] : p == OHF_P_NB_LAMBDA ? &g->nb_lambda[s][c] : &g->weight[s][c];
    double oldValue = *slot;
    *slot = value;
    double diffRatio = 1.0e-4 < oldValue ? fabs(value / oldValue - 1.0) : 0.0;
    return diffRatio < tol;
}

/* hmm_utils.c:1885-1900: theta, lambda, weight through EmissionDistSeries_estimateOneParameterType (:1817-1858) */
bool ohf_nb_estimate(const ohf_model *m, ohf_region *g, double tol) {
    bool converged = true;
    for (int p = 0; p < 3; p++) {
        double bnum = 0.0, bden = 0.0;
        for (int s = 0; s < OHF_NSTATES; s++)
            for (int c = 0; c < m->ncomp[s]; c++) {
                double factor = nb_binding_coef(s, p, c);
                if (0.0 < factor) { bnum += g->est[s].num[p][c] / factor; bden += g->est[s].den[p][c]; }
            }
        double boundCount = bden;
        double boundEstimation = (bden == 0) ? 0.0 : bnum / bden;
        for (int s = 0; s < OHF_NSTATES; s++)
            for (int c = 0; c < m->ncomp[s]; c++) {
                double factor = nb_binding_coef(s, p, c);
                double estimation, count;
                if (0.0 < factor) { estimation = boundEstimation * factor; count = boundCount; }
                else {
                    count = g->est[s].den[p][c];
                    estimation = (count == 0) ? 0.0 : g->est[s].num[p][c] / g->est[s].den[p][c];
                }
                if (10 < count) converged &= nb_update_param(g, s, p, c, estimation, tol);    /* MIN_COUNT_FOR_PARAMETER_UPDATE */
            }
    }
    return converged;
}
