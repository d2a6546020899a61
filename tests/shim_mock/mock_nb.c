/* tests/shim_mock/mock_nb.c — TEST INFRASTRUCTURE.  Bodies for the two NegativeBinomial functions of the reference that
 * integration/hmm_hip_shim.c calls (hmm_utils.h:255, 275), on top of the oracle's restatement of them (oracle/ohf_nb.c, linked
 * as liboracle_hf.so), plus the digamma table the reference keeps in the struct (hmm_utils.c:394-408) for the mock driver. */
#include "hmm.h"
#include "ohf.h"
#include <stdio.h>
#include <stdlib.h>

double NegativeBinomial_getR(double theta, double lambda) { return ohf_nb_r(theta, lambda); }

static void as_region(const NegativeBinomial *nb, ohf_region *g) {
    for (int c = 0; c < nb->numberOfComps; c++) { g->theta[0][c] = nb->theta[c]; g->nb_lambda[0][c] = nb->lambda[c]; g->weight[0][c] = nb->weights[c]; }
}

double *NegativeBinomial_getComponentProbs(NegativeBinomial *nb, uint8_t x) {
    static ohf_region g;
    as_region(nb, &g);
    double *probs = malloc(nb->numberOfComps * sizeof(double));
    if (ohf_nb_comp_probs(&g, 0, nb->numberOfComps, x, probs) < 0) { fprintf(stderr, "prob is NAN\n"); exit(EXIT_FAILURE); }   /* hmm_utils.c:511-514 */
    return probs;
}

void mock_nb_fill_digamma_table(NegativeBinomial *nb) {
    static ohf_region g;
    static double table[OHF_MAXCOMP][OHF_MAX_COVERAGE_VALUE + 1];
    as_region(nb, &g);
    ohf_nb_digamma_table(&g, 0, nb->numberOfComps, table);
    nb->digammaTable = calloc(nb->numberOfComps, sizeof(double *));
    for (int c = 0; c < nb->numberOfComps; c++) {
        nb->digammaTable[c] = malloc((OHF_MAX_COVERAGE_VALUE + 1) * sizeof(double));
        for (int x = 0; x <= OHF_MAX_COVERAGE_VALUE; x++) nb->digammaTable[c][x] = table[c][x];
    }
}
