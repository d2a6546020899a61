/*
 * ohf_main.c — ORACLE command line (test infrastructure, not product code).
 * Mirrors the option handling of /root/reference/programs/src/hmm_flagger.c:611-1077 for the
 * options the parity tests use; input is the reference's `.bin` chunk format (or .cov/.cov.gz
 * when built with the oracle loader).
 */
#include "ohf.h"
#include <getopt.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/time.h>

ohf_chunks *ohf_read_cov(const char *path, int chunk_len, int window_len); /* ohf_cov.c */

static struct option long_options[] = { /* hmm_flagger.c:578-608 */
    {"input", required_argument, NULL, 'i'}, {"preset", required_argument, NULL, 'x'},
    {"iterations", required_argument, NULL, 'n'}, {"convergenceTol", required_argument, NULL, 't'},
    {"disableAdjustContigEnds", no_argument, NULL, 'e'}, {"minReadFractionAtEnds", required_argument, NULL, 'f'},
    {"modelType", required_argument, NULL, 'm'}, {"maxHighMapqRatio", required_argument, NULL, 'q'},
    {"minHighMapqRatio", required_argument, NULL, 'Q'}, {"chunkLen", required_argument, NULL, 'C'},
    {"windowLen", required_argument, NULL, 'W'}, {"threads", required_argument, NULL, '@'},
    {"collapsedComps", required_argument, NULL, 'p'}, {"alphaTsv", required_argument, NULL, 'A'},
    {"writeParameterStatsPerIteration", no_argument, NULL, 'w'}, {"writePosteriorProbs", no_argument, NULL, 'P'},
    {"outputDir", required_argument, NULL, 'o'}, {"trackName", required_argument, NULL, 'N'},
    {"dumpBin", no_argument, NULL, 'B'}, {"minimumLengths", required_argument, NULL, 'M'},
    {"accelerate", no_argument, NULL, 's'},
    {NULL, 0, NULL, 0}};

static const char *file_ext(const char *p) { /* common.c:51-66 */
    int len = (int) strlen(p), i = len - 1;
    for (; 0 <= i; i--)
        if (p[i] == '.')
            if (strcmp(p + i, ".gz") != 0 && strcmp(p + i, ".tar") != 0 && strcmp(p + i, ".tar.gz") != 0 &&
                strcmp(p + i, ".zip") != 0) break;
    return p + i + 1;
}

static double now(void) { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + 1e-6 * tv.tv_usec; }

int main(int argc, char **argv) {
    const char *trackName = "final_hmm_flagger", *preset = "hifi", *inputPath = NULL, *alphaTsv = NULL, *outDir = NULL;
    int iterations = 100, collapsed = -1, chunkLen = 20000000, windowLen = -1, threads = 4, modelType = -1;
    double tol = 0.001, maxHighMapq = 0.25, minHighMapq = 0.75, minReadFrac = -1.0;
    bool adjust = true, wparams = false, wpost = false, dumpBin = false, accel = false;
    int minLen[4] = {0, 0, 0, 0};
    int c;
    while (~(c = getopt_long(argc, argv, "i:x:f:en:t:m:q:Q:C:W:@:p:A:wPo:BN:M:s", long_options, NULL))) {
        switch (c) {
            case 'i': inputPath = optarg; break;
            case 'x': preset = optarg; break;
            case 'n': iterations = atoi(optarg); break;
            case 't': tol = atof(optarg); break;
            case 'e': adjust = false; break;
            case 'f': minReadFrac = atof(optarg); break;
            case 'm':
                if (!strcmp(optarg, "gaussian")) modelType = OHF_MODEL_GAUSSIAN;
                else if (!strcmp(optarg, "trunc_exp_gaussian") || !strcmp(optarg, "truncated_exponential_gaussian"))
                    modelType = OHF_MODEL_TRUNC_EXP_GAUSSIAN;
                else if (!strcmp(optarg, "nb") || !strcmp(optarg, "negative_binomial")) /* hmm_utils.c:18-29 */
                    modelType = OHF_MODEL_NEGATIVE_BINOMIAL;
                else { fprintf(stderr, "oracle: unsupported model type %s\n", optarg); return 1; }
                break;
            case 'q': maxHighMapq = atof(optarg); break;
            case 'Q': minHighMapq = atof(optarg); break;
            case 'C': chunkLen = atoi(optarg); break;
            case 'W': windowLen = atoi(optarg); break;
            case '@': threads = atoi(optarg); break;
            case 'p': collapsed = atoi(optarg); break;
            case 'A': alphaTsv = optarg; break;
            case 'w': wparams = true; break;
            case 'P': wpost = true; break;
            case 'o': outDir = optarg; break;
            case 'B': dumpBin = true; break;
            case 's': accel = true; break;
            case 'N': trackName = optarg; break;
            case 'M': {
                int a, b, d;
                if (sscanf(optarg, "%d,%d,%d", &a, &b, &d) != 3) { fprintf(stderr, "bad --minimumLengths\n"); return 1; }
                minLen[0] = a; minLen[1] = b; minLen[3] = d; /* hmm_flagger.c:745-747 */
                break;
            }
            default: fprintf(stderr, "oracle: undefined option\n"); return 1;
        }
    }
    if (!inputPath || !outDir) { fprintf(stderr, "oracle: -i and -o are required\n"); return 1; }
    struct stat sb;
    if (stat(outDir, &sb) != 0 || !S_ISDIR(sb.st_mode)) { fprintf(stderr, "Error: Output directory %s does not exist!\n", outDir); return 1; }
    /* presets: hmm_flagger.c:17-58, 945-956; preset alpha arrays are `int` => all zero (:21,36,50) */
    double alpha[4][4]; memset(alpha, 0, sizeof(alpha));
    if (alphaTsv && ohf_read_alpha_tsv(alphaTsv, alpha) != 0) { fprintf(stderr, "oracle: bad alpha tsv\n"); return 1; }
    int presetW; double presetF;
    if (!strcmp(preset, "hifi")) { presetW = 16000; presetF = 0.95; }
    else if (!strcmp(preset, "ont-r9")) { presetW = 16000; presetF = 1.0; }
    else if (!strcmp(preset, "ont-r10")) { presetW = 8000; presetF = 0.8; }
    else { fprintf(stderr, "Error: preset can be one of hifi, ont-r9, ont-r10.\n"); return 1; }
    if (minReadFrac < 0.0 && adjust) minReadFrac = presetF;
    if (windowLen < 0) windowLen = presetW;
    if (modelType < 0) modelType = OHF_MODEL_TRUNC_EXP_GAUSSIAN;

    const char *ext = file_ext(inputPath);
    ohf_chunks *cc = NULL;
    if (!strcmp(ext, "bin")) cc = ohf_read_bin(inputPath);
    else if (!strcmp(ext, "cov") || !strcmp(ext, "cov.gz")) cc = ohf_read_cov(inputPath, chunkLen, windowLen);
    else { fprintf(stderr, "Error: input file should either cov/cov.gz or a binary file\n"); return 1; }
    if (!cc) { fprintf(stderr, "oracle: cannot read %s\n", inputPath); return 1; }
    if (dumpBin) {
        char p[2200];
        snprintf(p, sizeof(p), "%s/chunks.c_%d.w_%d.bin", outDir, cc->chunk_len, cc->window_len);
        ohf_write_bin(cc, p);
    }
    if (collapsed == -1) collapsed = ohf_best_collapsed_comps(cc);
    ohf_model *m = ohf_model_create(modelType, collapsed, cc->region_coverages, cc->n_regions, cc->start_only,
                                    cc->avg_alignment_len, cc->window_len, alpha, maxHighMapq, minHighMapq);
    if (!m) { fprintf(stderr, "oracle: cannot create model\n"); return 1; }
    ohf_run_opts o = { adjust, minReadFrac, cc->avg_alignment_len, threads };
    ohf_em_opts eo = { iterations, tol, wparams, wpost, outDir, accel };
    double t0 = now();
    int passes = ohf_run_em(cc, m, &o, &eo, NULL, 0);
    double t1 = now();
    if (passes < 0) {
        fprintf(stderr, passes == -1 ? "scale is very low!\n" : "oracle: E-step failed (%d)\n", passes);
        return 1;
    }
    long nwin = 0;
    for (int k = 0; k < cc->n_chunks; k++) nwin += cc->chunks[k].n;
    fprintf(stderr, "oracle: %d passes over %ld windows in %.3f s (%.1f windows/s, %d threads)\n",
            passes, nwin, t1 - t0, (double) nwin * passes / (t1 - t0), threads);
    char bed[2200];
    snprintf(bed, sizeof(bed), "%s/final_flagger_prediction.bed", outDir);
    ohf_write_final_bed(cc, bed, trackName, minLen);
    ohf_model_destroy(m);
    ohf_chunks_destroy(cc);
    return 0;
}
