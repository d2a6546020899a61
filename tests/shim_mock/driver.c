/* tests/shim_mock/driver.c — TEST INFRASTRUCTURE.  Builds the reference-shaped objects (tests/shim_mock/hmm.h) from a flat dump
 * written by the test, calls the three functions of integration/hmm_hip_shim.c the way hmm_flagger.c does (:344, :464, :272)
 * and writes back what the reference would read next: model->loglikelihood, the estimator numerators / denominators, the
 * transition counts, the predictions stored in the Inference objects, and two posteriors.
 * dump (little endian): i32 C, R, K, modelType, windowLen, meanReadLen, adjust; f64 minFrac, maxMapq, minMapq, minClip;
 *   f64 alpha[16]; per chunk: i32 n, s, e, ctgLen; then u16 cov[N], mapq[N], clip[N]; u64 annot[N];
 *   per region: f64 trans[25], lambda, trunc, mean[4][16], var[4][16], weight[4][16] */
#include "hmm.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static MatrixDouble *mat(int n) {
    MatrixDouble *m = calloc(1, sizeof *m);
    m->dim1 = m->dim2 = n; m->data = calloc(n, sizeof(double *));
    for (int i = 0; i < n; i++) m->data[i] = calloc(n, sizeof(double));
    return m;
}
static ParameterEstimator *est(int n) {
    ParameterEstimator *p = calloc(1, sizeof *p);
    p->numberOfComps = n; p->numeratorPerComp = calloc(n, 8); p->denominatorPerComp = calloc(n, 8);
    return p;
}
void mock_nb_fill_digamma_table(NegativeBinomial *nb);      /* mock_nb.c */
#define RD(ptr, n) do { if (fread(ptr, 1, (n), f) != (size_t) (n)) { fprintf(stderr, "short dump\n"); return 2; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[7]; double d[4], alpha[16];
    RD(h, sizeof h); RD(d, sizeof d); RD(alpha, sizeof alpha);
    const int C = h[0], R = h[1], K = h[2];
    HMM model; memset(&model, 0, sizeof model);
    model.numberOfRegions = R; model.numberOfStates = 4; model.maxNumberOfComps = K; model.modelType = (ModelType) h[3];
    model.alpha = mat(4);
    for (int i = 0; i < 16; i++) model.alpha->data[i / 4][i % 4] = alpha[i];
    stList list; list.n = C; list.items = calloc(C, sizeof(void *));
    int64_t N = 0;
    for (int c = 0; c < C; c++) {
        int32_t g[4]; RD(g, sizeof g);
        EM *em = calloc(1, sizeof *em);
        em->chunk = calloc(1, sizeof(Chunk));
        em->seqLen = g[0]; em->chunk->s = g[1]; em->chunk->e = g[2]; em->chunk->ctgLen = g[3]; em->chunk->windowLen = h[4];
        em->meanReadLength = h[5]; em->adjustContigEnds = h[6] != 0; em->minReadFractionAtEnds = d[0];
        em->coverageInfoSeq = calloc(g[0], sizeof(CoverageInfo *));
        for (int i = 0; i < g[0]; i++) { em->coverageInfoSeq[i] = calloc(1, sizeof(CoverageInfo)); em->coverageInfoSeq[i]->data = calloc(1, sizeof(Inference)); }
        list.items[c] = em; N += g[0];
    }
    uint16_t *cov = malloc(2 * N), *mapq = malloc(2 * N), *clip = malloc(2 * N); uint64_t *annot = malloc(8 * N);
    RD(cov, 2 * N); RD(mapq, 2 * N); RD(clip, 2 * N); RD(annot, 8 * N);
    for (int64_t t = 0, c = 0, i = 0; t < N; t++, i++) {
        while (i == ((EM *) list.items[c])->seqLen) { c++; i = 0; }
        CoverageInfo *ci = ((EM *) list.items[c])->coverageInfoSeq[i];
        ci->coverage = cov[t]; ci->coverage_high_mapq = mapq[t]; ci->coverage_high_clip = clip[t]; ci->annotation_flag = annot[t];
    }
    model.transitionPerRegion = calloc(R, sizeof(Transition *));
    model.emissionDistSeriesPerRegion = calloc(R, sizeof(EmissionDistSeries *));
    for (int r = 0; r < R; r++) {
        double p[27 + 3 * 64]; RD(p, sizeof p);
        Transition *tr = calloc(1, sizeof *tr);
        tr->matrix = mat(5); tr->transitionCountData = calloc(1, sizeof(TransitionCountData)); tr->transitionCountData->countMatrix = mat(5);
        tr->requirements = calloc(1, sizeof(TransitionRequirements));
        tr->requirements->minHighlyClippedRatio = d[3]; tr->requirements->maxHighMapqRatio = d[1]; tr->requirements->minHighMapqRatio = d[2];
        for (int i = 0; i < 25; i++) tr->matrix->data[i / 5][i % 5] = p[i];
        model.transitionPerRegion[r] = tr;
        EmissionDistSeries *eds = calloc(1, sizeof *eds);
        eds->numberOfDists = 4; eds->emissionDists = calloc(4, sizeof(EmissionDist *));
        for (int s = 0; s < 4; s++) {
            EmissionDist *e = calloc(1, sizeof *e);
            const int nc = s == 3 ? K : 1;
            if (model.modelType == MODEL_NEGATIVE_BINOMIAL) {    /* theta / lambda / weights travel in the mean / var / weight slots */
                NegativeBinomial *nb = calloc(1, sizeof *nb);
                nb->numberOfComps = nc; nb->theta = calloc(nc, 8); nb->lambda = calloc(nc, 8); nb->weights = calloc(nc, 8);
                for (int c = 0; c < nc; c++) { nb->theta[c] = p[27 + s * 16 + c]; nb->lambda[c] = p[27 + 64 + s * 16 + c]; nb->weights[c] = p[27 + 128 + s * 16 + c]; }
                nb->thetaEstimator = est(nc); nb->lambdaEstimator = est(nc); nb->weightsEstimator = est(nc);
                mock_nb_fill_digamma_table(nb);
                e->dist = nb; e->distType = DIST_NEGATIVE_BINOMIAL;
            } else if (s == 0 && model.modelType == MODEL_TRUNC_EXP_GAUSSIAN) {
                TruncExponential *te = calloc(1, sizeof *te);
                te->lambda = p[25]; te->truncPoint = p[26]; te->lambdaEstimator = est(1);
                e->dist = te; e->distType = DIST_TRUNC_EXPONENTIAL;
            } else {
                Gaussian *g = calloc(1, sizeof *g);
                g->numberOfComps = nc; g->mean = calloc(nc, 8); g->var = calloc(nc, 8); g->weights = calloc(nc, 8);
                for (int c = 0; c < nc; c++) { g->mean[c] = p[27 + s * 16 + c]; g->var[c] = p[27 + 64 + s * 16 + c]; g->weights[c] = p[27 + 128 + s * 16 + c]; }
                g->meanEstimator = est(nc); g->varEstimator = est(nc); g->weightsEstimator = est(nc);
                e->dist = g; e->distType = DIST_GAUSSIAN;
            }
            eds->emissionDists[s] = e;
        }
        model.emissionDistSeriesPerRegion[r] = eds;
    }
    fclose(f);
    EM_runForwardForList(&list, &model, 4);                     /* as the SQUAREM line search does, hmm.c:900 */
    const double ll_forward = model.loglikelihood;
    EM_runOneIterationForList(&list, &model, 4);                /* hmm_flagger.c:344 */
    FILE *o = fopen(argv[2], "wb");
    fwrite(&ll_forward, 8, 1, o); fwrite(&model.loglikelihood, 8, 1, o);
    for (int r = 0; r < R; r++) {                               /* in the layout of include/hmm_flagger_hip.h, K slots per row */
        for (int s = 0; s < 4; s++) {
            EmissionDist *e = model.emissionDistSeriesPerRegion[r]->emissionDists[s];
            ParameterEstimator *pe[3] = {NULL, NULL, NULL};
            if (e->distType == DIST_TRUNC_EXPONENTIAL) pe[0] = ((TruncExponential *) e->dist)->lambdaEstimator;
            else if (e->distType == DIST_NEGATIVE_BINOMIAL) { NegativeBinomial *nb = e->dist; pe[0] = nb->thetaEstimator; pe[1] = nb->lambdaEstimator; pe[2] = nb->weightsEstimator; }
            else { Gaussian *g = e->dist; pe[0] = g->meanEstimator; pe[1] = g->varEstimator; pe[2] = g->weightsEstimator; }
            for (int q = 0; q < 3; q++)
                for (int k = 0; k < 2; k++)
                    for (int c = 0; c < K; c++) {
                        double v = 0.0;
                        if (pe[q] && c < pe[q]->numberOfComps) v = k == 0 ? pe[q]->numeratorPerComp[c] : pe[q]->denominatorPerComp[c];
                        fwrite(&v, 8, 1, o);
                    }
        }
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) fwrite(&model.transitionPerRegion[r]->transitionCountData->countMatrix->data[i][j], 8, 1, o);
    }
    for (int c = 0; c < C; c++) {
        EM *em = list.items[c];
        for (int i = 0; i < em->seqLen; i++) fwrite(&((Inference *) em->coverageInfoSeq[i]->data)->prediction, 1, 1, o);
    }
    double *p0 = EM_getPosterior(list.items[0], 0), *p1 = EM_getPosterior(list.items[C - 1], ((EM *) list.items[C - 1])->seqLen - 1);
    fwrite(p0, 8, 4, o); fwrite(p1, 8, 4, o);
    fclose(o);
    return 0;
}
