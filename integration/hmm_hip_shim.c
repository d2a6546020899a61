/* hmm_hip_shim.c — what a maintainer of mobinasri/flagger adds to re-point hmm_flagger's E-step at the MI355X library.
 * Goes in programs/submodules/hmm/ next to hmm.c, whose EM_runOneIterationForList / EM_runForwardForList / EM_getPosterior
 * (hmm.c:739, 790, 671) are renamed or guarded by -DFLAGGER_HIP; built with
 *     gcc ... -I<repo>/include -L<repo>/flagger_amd/csrc -lhmmflagger_hip
 * Only reference declarations (hmm.h) and the C ABI (include/hmm_flagger_hip.h) are used.  INTEGRATION.md walks through it;
 * tests/test_shim_cpu.py / tests/test_cli_gpu.py compile THIS file against tests/shim_mock/hmm.h (the reference's struct
 * fields, nothing else) and drive it end to end. */
#include "hmm.h"
#include "hmm_flagger_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static hf_ctx *g_ctx = NULL;             /* windows stay resident in HBM for the whole run */
static int64_t *g_off; static int g_K;
static stList *g_list = NULL;            /* the EM list the context was built from (EM_getPosterior finds its chunk in it) */

static void flatten_and_create(stList *emList, HMM *model) {
    int C = stList_length(emList); int64_t N = 0;
    g_list = emList;
    g_off = malloc((C + 1) * sizeof(int64_t));
    for (int c = 0; c < C; c++) { g_off[c] = N; N += ((EM *) stList_get(emList, c))->seqLen; } g_off[C] = N;
    uint16_t *cov = malloc(2 * N), *mapq = malloc(2 * N), *clip = malloc(2 * N); uint64_t *annot = malloc(8 * N);
    int32_t *cs = malloc(4 * C), *ce = malloc(4 * C), *cl = malloc(4 * C);
    EM *em0 = stList_get(emList, 0);
    for (int c = 0; c < C; c++) {
        EM *em = stList_get(emList, c);
        cs[c] = em->chunk->s; ce[c] = em->chunk->e; cl[c] = em->chunk->ctgLen;          /* chunk.h:16-23 */
        for (int i = 0; i < em->seqLen; i++) {                                           /* ptBlock.h:79-92 */
            CoverageInfo *ci = em->coverageInfoSeq[i]; int64_t t = g_off[c] + i;
            cov[t] = ci->coverage; mapq[t] = ci->coverage_high_mapq; clip[t] = ci->coverage_high_clip;
            annot[t] = ci->annotation_flag;
        }
    }
    Transition *tr = model->transitionPerRegion[0];
    hf_windows w = { N, C, g_off, cov, mapq, clip, annot, cs, ce, cl, em0->chunk->windowLen, em0->meanReadLength,
                     em0->adjustContigEnds, em0->minReadFractionAtEnds, tr->requirements->maxHighMapqRatio,
                     tr->requirements->minHighMapqRatio, tr->requirements->minHighlyClippedRatio };
    g_K = model->maxNumberOfComps;
    if (hf_create(&w, model->numberOfRegions, g_K, /*device*/0, HF_ALGO_SCAN, &g_ctx) != HF_OK) {
        fprintf(stderr, "[hmm_flagger] %s\n", hf_last_error()); exit(EXIT_FAILURE);     /* no CPU fallback */
    }
    free(cov); free(mapq); free(clip); free(annot); free(cs); free(ce); free(cl);
}

/* negative_binomial: every emission quantity depends on the coverage value alone, so the library takes per-x tables that the
 * HOST fills with the reference's own functions (hf_params.nb_*, include/hmm_flagger_hip.h:84-95) — here literally
 * NegativeBinomial_getComponentProbs / NegativeBinomial_getR (hmm_utils.c:497-520, 458-461) and nb->digammaTable (:394-408) */
#define NB_NX (HF_NB_MAX_COVERAGE + 1)
static double *g_nbE, *g_nbP, *g_nbDig, *g_nbR, *g_nbBeta;

static void fill_nb_tables(HMM *model, hf_params *p, int K) {
    const int R = model->numberOfRegions;
    if (!g_nbE) {
        g_nbE = calloc((size_t) R * 4 * NB_NX, 8); g_nbP = calloc((size_t) R * 4 * K * NB_NX, 8);
        g_nbDig = calloc((size_t) R * 4 * K * NB_NX, 8); g_nbR = calloc((size_t) R * 4 * K, 8); g_nbBeta = calloc((size_t) R * 4 * K, 8);
    }
    for (int r = 0; r < R; r++)
        for (int s = 0; s < 4; s++) {
            NegativeBinomial *nb = model->emissionDistSeriesPerRegion[r]->emissionDists[s]->dist;
            for (int c = 0; c < nb->numberOfComps; c++) {
                const size_t pc = ((size_t) r * 4 + s) * K + c;
                g_nbR[pc] = NegativeBinomial_getR(nb->theta[c], nb->lambda[c]);
                g_nbBeta[pc] = -1 * nb->theta[c] / (1 - nb->theta[c]) - 1 / log(nb->theta[c]);           /* hmm_utils.c:547 */
                for (int x = 0; x < NB_NX; x++) g_nbDig[pc * NB_NX + x] = nb->digammaTable[c][x];
            }
            for (int x = 0; x < NB_NX; x++) {
                double *probs = NegativeBinomial_getComponentProbs(nb, (uint8_t) x), tot = 0.0;
                for (int c = 0; c < nb->numberOfComps; c++) {                                           /* Double_sum1DArray: index order */
                    g_nbP[(((size_t) r * 4 + s) * K + c) * NB_NX + x] = probs[c]; tot += probs[c];
                }
                g_nbE[((size_t) r * 4 + s) * NB_NX + x] = tot;                                           /* NegativeBinomial_getProb */
                free(probs);
            }
        }
    p->nb_E = g_nbE; p->nb_P = g_nbP; p->nb_dig = g_nbDig; p->nb_r = g_nbR; p->nb_beta = g_nbBeta;
    p->nb_max_x = 0;                                              /* the tables are filled for every x = 0..250 */
}

static void pack_params(HMM *model, hf_params *p, double *trans, double *lam, double *trunc,
                        double *mean, double *var, double *weight) {
    memset(p, 0, sizeof *p);
    switch (model->modelType) {
        case MODEL_GAUSSIAN: p->model_type = HF_MODEL_GAUSSIAN; break;
        case MODEL_TRUNC_EXP_GAUSSIAN: p->model_type = HF_MODEL_TRUNC_EXP_GAUSSIAN; break;
        case MODEL_NEGATIVE_BINOMIAL: p->model_type = HF_MODEL_NEGATIVE_BINOMIAL; break;
        default: fprintf(stderr, "[hmm_flagger] model type %d has no HIP E-step\n", (int) model->modelType); exit(EXIT_FAILURE);
    }
    p->n_regions = model->numberOfRegions;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) p->alpha[i][j] = model->alpha->data[i][j];
    for (int r = 0; r < model->numberOfRegions; r++) {
        for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++)
            trans[r * 25 + i * 5 + j] = model->transitionPerRegion[r]->matrix->data[i][j];
        EmissionDistSeries *eds = model->emissionDistSeriesPerRegion[r];
        for (int s = 0; s < 4; s++) {
            EmissionDist *d = eds->emissionDists[s];
            if (d->distType == DIST_TRUNC_EXPONENTIAL) {
                lam[r] = ((TruncExponential *) d->dist)->lambda; trunc[r] = ((TruncExponential *) d->dist)->truncPoint;
                p->ncomp[s] = 1;
            } else if (d->distType == DIST_NEGATIVE_BINOMIAL) {          /* theta / lambda / weights in the mean / var / weight slots */
                NegativeBinomial *nb = d->dist; p->ncomp[s] = nb->numberOfComps;
                for (int c = 0; c < nb->numberOfComps; c++) {
                    size_t o = ((size_t) r * 4 + s) * HF_MAXCOMP + c;
                    mean[o] = nb->theta[c]; var[o] = nb->lambda[c]; weight[o] = nb->weights[c];
                }
            } else if (d->distType != DIST_GAUSSIAN) {
                fprintf(stderr, "[hmm_flagger] emission distribution %d has no HIP E-step\n", (int) d->distType); exit(EXIT_FAILURE);
            } else {
                Gaussian *g = d->dist; p->ncomp[s] = g->numberOfComps;
                for (int c = 0; c < g->numberOfComps; c++) {
                    size_t o = ((size_t) r * 4 + s) * HF_MAXCOMP + c;
                    mean[o] = g->mean[c]; var[o] = g->var[c]; weight[o] = g->weights[c];
                }
            }
        }
    }
    p->trans = trans; p->lambda = lam; p->trunc_point = trunc; p->mean = mean; p->var = var; p->weight = weight;
    if (model->modelType == MODEL_NEGATIVE_BINOMIAL) fill_nb_tables(model, p, model->maxNumberOfComps);
}

static void run(stList *emList, HMM *model, int mode) {
    if (!g_ctx) flatten_and_create(emList, model);
    int R = model->numberOfRegions, K = g_K;
    double *trans = calloc(R * 25, 8), *lam = calloc(R, 8), *trunc = calloc(R, 8);
    double *mean = calloc((size_t) R * 4 * HF_MAXCOMP, 8), *var = calloc((size_t) R * 4 * HF_MAXCOMP, 8),
           *weight = calloc((size_t) R * 4 * HF_MAXCOMP, 8), *st = calloc(hf_stats_len(R, K), 8);
    hf_params p; pack_params(model, &p, trans, lam, trunc, mean, var, weight);
    int rc = hf_estep(g_ctx, &p, mode, NULL);
    if (rc == HF_OK) rc = hf_finish(g_ctx, st, NULL);
    if (rc == HF_E_SCALE) { fprintf(stderr, "scale is very low!\n"); exit(EXIT_FAILURE); }        /* hmm.c:412-415 */
    if (rc == HF_E_NAN)   { fprintf(stderr, "[Error] prob is NAN\n"); exit(EXIT_FAILURE); }       /* hmm_utils.c:782-786 */
    if (rc != HF_OK)      { fprintf(stderr, "[hmm_flagger] %s\n", hf_last_error()); exit(EXIT_FAILURE); }
    model->loglikelihood = st[0];                                                                  /* hmm.c:761 */
    if (mode == HF_MODE_FULL) {
        for (int r = 0; r < R; r++) {            /* the model's (already reset) estimators := the statistics vector */
            const double *b = st + 1 + r * hf_region_stride(K);
            EmissionDistSeries *eds = model->emissionDistSeriesPerRegion[r];
            for (int s = 0; s < 4; s++) {
                EmissionDist *d = eds->emissionDists[s];
                ParameterEstimator *pe[3]; int np = 3;
                if (d->distType == DIST_TRUNC_EXPONENTIAL) { pe[0] = ((TruncExponential *) d->dist)->lambdaEstimator; np = 1; }
                else if (d->distType == DIST_NEGATIVE_BINOMIAL) {        /* parameter slots 0 / 1 / 2 = theta / lambda / weights */
                    NegativeBinomial *nb = d->dist; pe[0] = nb->thetaEstimator; pe[1] = nb->lambdaEstimator; pe[2] = nb->weightsEstimator;
                } else { Gaussian *g = d->dist; pe[0] = g->meanEstimator; pe[1] = g->varEstimator; pe[2] = g->weightsEstimator; }
                for (int q = 0; q < np; q++) for (int c = 0; c < pe[q]->numberOfComps; c++) {
                    pe[q]->numeratorPerComp[c]   += b[((s * 3 + q) * 2 + 0) * K + c];
                    pe[q]->denominatorPerComp[c] += b[((s * 3 + q) * 2 + 1) * K + c];
                }
            }
            MatrixDouble *cm = model->transitionPerRegion[r]->transitionCountData->countMatrix;
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) cm->data[i][j] += b[24 * K + i * 4 + j];
        }
        int8_t *lab = malloc(hf_n_windows(g_ctx)); hf_get_labels(g_ctx, lab);                     /* hmm.c:730-736 */
        for (int c = 0; c < stList_length(emList); c++) {
            EM *em = stList_get(emList, c);
            for (int i = 0; i < em->seqLen; i++)
                if (em->coverageInfoSeq[i]->data) ((Inference *) em->coverageInfoSeq[i]->data)->prediction = lab[g_off[c] + i];
        }
        free(lab);
    }
    free(trans); free(lam); free(trunc); free(mean); free(var); free(weight); free(st);
}

void EM_runOneIterationForList(stList *emList, HMM *model, int threads) { (void) threads; run(emList, model, HF_MODE_FULL); }
void EM_runForwardForList(stList *emList, HMM *model, int threads)      { (void) threads; run(emList, model, HF_MODE_FORWARD_ONLY); }

/* -P / writePosteriorIntoBED (hmm_flagger.c:272-273): EM_getPosterior(em, i) -> */
double *EM_getPosterior(EM *em, int pos) {
    double *post = malloc(4 * sizeof(double));
    int c = 0;
    while (c < stList_length(g_list) && stList_get(g_list, c) != (void *) em) c++;     /* the chunk of this EM object */
    if (c == stList_length(g_list) || hf_get_posterior(g_ctx, g_off[c] + pos, 1, post) != HF_OK) {
        fprintf(stderr, "[hmm_flagger] %s\n", hf_last_error()); exit(EXIT_FAILURE);
    }
    return post;
}
