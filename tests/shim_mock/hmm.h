/* tests/shim_mock/hmm.h — TEST INFRASTRUCTURE.  Mock of the declarations of mobinasri/flagger that integration/hmm_hip_shim.c
 * touches, so that the documented binding is compiled, linked and run: the struct FIELDS (names, types, order of the ones
 * used) are the reference's — hmm.h:14-25 (HMM), :65-80 (EM); hmm_utils.h:36-48 (DistType, ModelType), :86-89 (EmissionDist),
 * :93-105 (ParameterEstimator), :213-222 (NegativeBinomial), :312-320 (Gaussian), :393-397 (TruncExponential), :547-555 (EmissionDistSeries), :713-717
 * (TransitionRequirements), :768-771 (TransitionCountData), :826-834 (Transition); chunk.h:11-34 (Chunk); ptBlock.h:50-55
 * (Inference), :79-92 (CoverageInfo); data_types.h:26-30 (MatrixDouble) — and sonLib's stList is reduced to an array with
 * the two accessors the shim calls.  Nothing here is reference code: no function bodies, no algorithm (the two NegativeBinomial
 * functions the shim calls for --modelType negative_binomial get test-only bodies in mock_nb.c, on top of the oracle). */
#ifndef SHIM_MOCK_HMM_H
#define SHIM_MOCK_HMM_H
#include <stdbool.h>
#include <stdint.h>
#include <sys/types.h>

typedef struct stList { void **items; int n; } stList;
static inline int stList_length(stList *l) { return l->n; }
static inline void *stList_get(stList *l, int i) { return l->items[i]; }

typedef struct MatrixDouble { int dim1; int dim2; double **data; } MatrixDouble;
typedef enum DistType { DIST_TRUNC_EXPONENTIAL = 0, DIST_GAUSSIAN = 1, DIST_NEGATIVE_BINOMIAL = 2, DIST_UNDEFINED = 3 } DistType;
typedef enum ModelType { MODEL_TRUNC_EXP_GAUSSIAN = 0, MODEL_GAUSSIAN = 1, MODEL_NEGATIVE_BINOMIAL = 2, MODEL_UNDEFINED = 3 } ModelType;

typedef struct EmissionDist { void *dist; DistType distType; } EmissionDist;
typedef struct ParameterEstimator { double *numeratorPerComp; double *denominatorPerComp; int numberOfComps; EmissionDist *emissionDist; } ParameterEstimator;
typedef struct Gaussian {
    double *mean; double *var; double *weights;
    ParameterEstimator *meanEstimator; ParameterEstimator *varEstimator; ParameterEstimator *weightsEstimator;
    int numberOfComps;
} Gaussian;
typedef struct TruncExponential { double lambda; double truncPoint; ParameterEstimator *lambdaEstimator; } TruncExponential;
typedef struct NegativeBinomial {                               /* hmm_utils.h:213-222 */
    double *theta; double *lambda; double *weights;
    ParameterEstimator *lambdaEstimator; ParameterEstimator *thetaEstimator; ParameterEstimator *weightsEstimator;
    int numberOfComps; double **digammaTable;
} NegativeBinomial;
double NegativeBinomial_getR(double theta, double lambda);                       /* hmm_utils.h:255; mock body: mock_nb.c */
double *NegativeBinomial_getComponentProbs(NegativeBinomial *nb, uint8_t x);     /* hmm_utils.h:275 */
typedef struct EmissionDistSeries {
    EmissionDist **emissionDists; void **countDataPerDist; void **parameterBindingPerDist;
    int numberOfDists; ModelType modelType; int numberOfCollapsedComps; bool excludeMisjoin;
} EmissionDistSeries;
typedef struct TransitionRequirements { double minHighlyClippedRatio; double maxHighMapqRatio; double minHighMapqRatio; } TransitionRequirements;
typedef struct TransitionCountData { MatrixDouble *countMatrix; MatrixDouble *pseudoCountMatrix; int numberOfStates; } TransitionCountData;
typedef struct Transition {
    MatrixDouble *matrix; TransitionCountData *transitionCountData; int numberOfStates; void *validityFunctions;
    int numberOfValidityFunctions; TransitionRequirements *requirements; double terminationProb;
} Transition;

typedef struct Inference { int8_t truth; int8_t prediction; bool isTruthAvailableInFile; bool isPredictionAvailableInFile; } Inference;
typedef struct CoverageInfo {
    uint64_t annotation_flag; u_int16_t coverage; u_int16_t coverage_high_mapq; u_int16_t coverage_high_clip;
    void *data; void (*destruct_data)(void *); void *(*copy_data)(void *); void (*extend_data)(void *, void *);
} CoverageInfo;
typedef struct Chunk {
    CoverageInfo **coverageInfoSeq; char ctg[200]; int ctgLen; int s; int e; int coverageInfoSeqLen; int chunkCanonicalLen;
    int coverageInfoMaxSeqSize; int windowLen;
} Chunk;

typedef struct HMM {
    EmissionDistSeries **emissionDistSeriesPerRegion; Transition **transitionPerRegion; ModelType modelType; MatrixDouble *alpha;
    int numberOfRegions; int numberOfStates; int maxNumberOfComps; bool excludeMisjoin; double loglikelihood;
} HMM;
typedef struct EM {
    CoverageInfo **coverageInfoSeq; Chunk *chunk; int meanReadLength; bool adjustContigEnds; double minReadFractionAtEnds; int seqLen;
    double **f; double **b; double px; double *scales; HMM *model; EmissionDistSeries **emissionDistSeriesPerRegion;
    Transition **transitionPerRegion; int numberOfRegions; double loglikelihood;
} EM;

void EM_runOneIterationForList(stList *emList, HMM *model, int threads);
void EM_runForwardForList(stList *emList, HMM *model, int threads);
double *EM_getPosterior(EM *em, int pos);
#endif
