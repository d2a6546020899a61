/*
 * ohf_run.c — ORACLE (test infrastructure, not product code).
 * EM outer loop, restated from runHMMFlagger (/root/reference/programs/src/hmm_flagger.c:285-488),
 * without --accelerate and without the summary tables.
 */
#include "ohf.h"
#include <stdlib.h>
#include <string.h>

static void write_params(const ohf_model *m, const char *out_dir, const char *suffix) { /* hmm_flagger.c:119-132 */
    char path[2200];
    snprintf(path, sizeof(path), "%s/transition_%s.tsv", out_dir, suffix);
    FILE *f = fopen(path, "w");
    if (f) { ohf_write_transition_tsv(m, f); fclose(f); }
    snprintf(path, sizeof(path), "%s/emission_%s.tsv", out_dir, suffix);
    f = fopen(path, "w");
    if (f) { ohf_write_emission_tsv(m, f); fclose(f); }
}

int ohf_run_em(ohf_chunks *cc, ohf_model *m, const ohf_run_opts *o, const ohf_em_opts *eo,
               double *ll_trace, int ll_cap) {
    FILE *llf = NULL;
    char path[2200];
    int passes = 0;
    if (eo->out_dir) {
        snprintf(path, sizeof(path), "%s/loglikelihood.tsv", eo->out_dir);
        llf = fopen(path, "w");
        if (!llf) return -10;
        fprintf(llf, "#Iteration\tEffective_Iteration\tLoglikelihood\n"); /* hmm_flagger.c:308 */
        write_params(m, eo->out_dir, "initial");                          /* :333 */
    }
    int iter = 1;
    bool converged = false;
    int st = 0;
    while (iter <= eo->iterations && converged == false) { /* :337 */
        st = ohf_run_iteration(cc, m, o, 0);               /* :344 */
        if (st) break;
        cc->prediction_available = true;                   /* :353-354 */
        cc->n_labels = 4;
        if (llf) fprintf(llf, "%d\t%d\t%.4f\n", iter - 1, eo->accelerate ? 3 * (iter - 1) : iter - 1, m->loglikelihood); /* :357 */
        if (ll_trace && passes < ll_cap) ll_trace[passes] = m->loglikelihood;
        passes++;
        if (eo->accelerate) {                              /* :382-416 */
            st = ohf_squarem_iteration(cc, m, o, eo->tol, NULL);
            if (st) break;
        }
        converged = ohf_estimate_parameters(m, eo->tol);   /* :419 */
        ohf_reset_estimators(m);                           /* :425 */
        if (eo->write_params_per_iter && eo->out_dir) {    /* :431-443 */
            char suffix[64];
            snprintf(suffix, sizeof(suffix), eo->accelerate ? "iteration_accelerated_%d" : "iteration_%d", iter);
            write_params(m, eo->out_dir, suffix);
        }
        iter += 1;
    }
    if (!st) {
        st = ohf_run_iteration(cc, m, o, 0);               /* :464 final inference */
        cc->prediction_available = true;
        cc->n_labels = 4;
        if (!st) {
            if (llf) fprintf(llf, "%d\t%d\t%.4f\n", iter - 1, eo->accelerate ? 3 * (iter - 1) : iter - 1, m->loglikelihood); /* :467 */
            if (ll_trace && passes < ll_cap) ll_trace[passes] = m->loglikelihood;
            passes++;
            if (eo->out_dir) {
                write_params(m, eo->out_dir, "final");     /* :475 */
                if (eo->write_posterior) {                 /* :483-485 */
                    snprintf(path, sizeof(path), "%s/posterior_prediction_final.bed", eo->out_dir);
                    ohf_write_posterior_bed(cc, path);
                }
            }
        }
    }
    if (llf) fclose(llf);
    return st ? st : passes;
}
