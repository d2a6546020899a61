/*
 * ohf_io.c — ORACLE (test infrastructure, not product code).
 * File formats either side of the hot path, restated from the reference
 * (citations: file:line under /root/reference/programs/).
 */
#include "ohf.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------- .bin (submodules/chunk/chunk.c:596-709 write, 713-828 read) ---------- */
static int rd(void *p, size_t sz, size_t n, FILE *f) { return fread(p, sz, n, f) == n ? 0 : -1; }

ohf_chunks *ohf_read_bin(const char *path) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    ohf_chunks *cc = calloc(1, sizeof(ohf_chunks));
    int bad = 0;
    bad |= rd(&cc->n_annotations, 4, 1, fp);
    if (bad || cc->n_annotations < 0 || cc->n_annotations > 64) { fclose(fp); free(cc); return NULL; }
    cc->annotation_names = calloc((size_t) cc->n_annotations + 1, sizeof(char *));
    for (int i = 0; i < cc->n_annotations; i++) {
        int32_t len = 0;
        bad |= rd(&len, 4, 1, fp);
        if (bad || len <= 0 || len > 4096) { bad = 1; break; }
        cc->annotation_names[i] = malloc((size_t) len);
        bad |= rd(cc->annotation_names[i], 1, (size_t) len, fp);
    }
    bad |= rd(&cc->n_regions, 4, 1, fp);
    if (!bad && (cc->n_regions < 0 || cc->n_regions > OHF_MAXREGIONS)) bad = 1;
    if (!bad) bad |= rd(cc->region_coverages, 4, (size_t) cc->n_regions, fp);
    bad |= rd(&cc->n_labels, 4, 1, fp);
    uint8_t b3[3] = {0, 0, 0};
    bad |= rd(b3, 1, 3, fp);
    cc->truth_available = b3[0]; cc->prediction_available = b3[1]; cc->start_only = b3[2];
    bad |= rd(&cc->avg_alignment_len, 4, 1, fp);
    bad |= rd(&cc->chunk_len, 4, 1, fp);
    bad |= rd(&cc->window_len, 4, 1, fp);
    int cap = 0;
    int32_t nameLen;
    while (!bad && fread(&nameLen, 4, 1, fp) == 1) {
        if (cc->n_chunks == cap) { cap = cap ? cap * 2 : 64; cc->chunks = realloc(cc->chunks, sizeof(ohf_chunk) * (size_t) cap); }
        ohf_chunk *ch = &cc->chunks[cc->n_chunks];
        memset(ch, 0, sizeof(*ch));
        if (nameLen <= 0 || nameLen > (int) sizeof(ch->ctg)) { bad = 1; break; }
        bad |= rd(ch->ctg, 1, (size_t) nameLen, fp);
        bad |= rd(&ch->ctg_len, 4, 1, fp);
        bad |= rd(&ch->s, 4, 1, fp);
        bad |= rd(&ch->e, 4, 1, fp);
        bad |= rd(&ch->n, 4, 1, fp);
        if (bad || ch->n < 0) { bad = 1; break; }
        size_t n = (size_t) ch->n;
        ch->cov = malloc(2 * n + 2); ch->mapq = malloc(2 * n + 2); ch->clip = malloc(2 * n + 2);
        ch->annot = malloc(8 * n + 8); ch->truth = malloc(n + 1); ch->prediction = malloc(n + 1);
        bad |= rd(ch->cov, 2, n, fp); bad |= rd(ch->mapq, 2, n, fp); bad |= rd(ch->clip, 2, n, fp);
        bad |= rd(ch->annot, 8, n, fp); bad |= rd(ch->truth, 1, n, fp); bad |= rd(ch->prediction, 1, n, fp);
        cc->n_chunks++;
    }
    fclose(fp);
    if (bad) { ohf_chunks_destroy(cc); return NULL; }
    return cc;
}

int ohf_write_bin(const ohf_chunks *cc, const char *path) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    fwrite(&cc->n_annotations, 4, 1, fp);
    for (int i = 0; i < cc->n_annotations; i++) {
        int32_t len = (int32_t) strlen(cc->annotation_names[i]) + 1;
        fwrite(&len, 4, 1, fp);
        fwrite(cc->annotation_names[i], 1, (size_t) len, fp);
    }
    fwrite(&cc->n_regions, 4, 1, fp);
    fwrite(cc->region_coverages, 4, (size_t) cc->n_regions, fp);
    fwrite(&cc->n_labels, 4, 1, fp);
    uint8_t b3[3] = { cc->truth_available, cc->prediction_available, cc->start_only };
    fwrite(b3, 1, 3, fp);
    fwrite(&cc->avg_alignment_len, 4, 1, fp);
    fwrite(&cc->chunk_len, 4, 1, fp);
    fwrite(&cc->window_len, 4, 1, fp);
    for (int c = 0; c < cc->n_chunks; c++) {
        const ohf_chunk *ch = &cc->chunks[c];
        int32_t nameLen = (int32_t) strlen(ch->ctg) + 1;
        fwrite(&nameLen, 4, 1, fp);
        fwrite(ch->ctg, 1, (size_t) nameLen, fp);
        fwrite(&ch->ctg_len, 4, 1, fp); fwrite(&ch->s, 4, 1, fp); fwrite(&ch->e, 4, 1, fp); fwrite(&ch->n, 4, 1, fp);
        size_t n = (size_t) ch->n;
        fwrite(ch->cov, 2, n, fp); fwrite(ch->mapq, 2, n, fp); fwrite(ch->clip, 2, n, fp);
        fwrite(ch->annot, 8, n, fp); fwrite(ch->truth, 1, n, fp); fwrite(ch->prediction, 1, n, fp);
    }
    fclose(fp);
    return 0;
}

void ohf_chunks_destroy(ohf_chunks *cc) {
    if (!cc) return;
    for (int c = 0; c < cc->n_chunks; c++) {
        ohf_chunk *ch = &cc->chunks[c];
        free(ch->cov); free(ch->mapq); free(ch->clip); free(ch->annot); free(ch->truth); free(ch->prediction);
        free(ch->f); free(ch->b); free(ch->scales);
    }
    free(cc->chunks);
    if (cc->annotation_names) for (int i = 0; i < cc->n_annotations; i++) free(cc->annotation_names[i]);
    free(cc->annotation_names);
    free(cc);
}

/* ---------- alpha TSV (src/hmm_flagger.c:491-515, submodules/data_types/data_types.c:490-518) ---------- */
int ohf_read_alpha_tsv(const char *path, double alpha[4][4]) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    memset(alpha, 0, sizeof(double) * 16);
    char *line = NULL; size_t cap = 0; int i = 0;
    while (getline(&line, &cap, f) != -1) {
        if (i >= 4) break;
        size_t L = strlen(line);
        if (L > 0) line[L - 1] = '\0'; /* data_types.c:507: drops the last character */
        int j = 0;
        char *save = NULL;
        for (char *tok = strtok_r(line, "\t", &save); tok && j < 4; tok = strtok_r(NULL, "\t", &save))
            alpha[i][j++] = atof(tok);
        i++;
    }
    free(line);
    fclose(f);
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++)
            if (1.0 < alpha[a][b] || alpha[a][b] < 0.0) return -2; /* hmm_flagger.c:503-513 */
    return 0;
}

/* ---------- parameter TSVs (submodules/hmm/hmm.c:137-239) ---------- */
static const char *STATE_NAMES[5] = {"Err", "Dup", "Hap", "Col", "Msj"}; /* hmm_utils.h:74 */

void ohf_write_transition_tsv(const ohf_model *m, FILE *f) {
    fprintf(f, "#Region\tState\tErr\tDup\tHap\tCol\tEnd\n");
    for (int r = 0; r < m->n_regions; r++)
        for (int p = 0; p < OHF_NSTATES + 1; p++) {
            fprintf(f, "%d\t%s", r, p < OHF_NSTATES ? STATE_NAMES[p] : "Start");
            for (int s = 0; s < OHF_NSTATES + 1; s++) fprintf(f, "\t%.5e", m->regions[r].trans[p][s]);
            fprintf(f, "\n");
        }
}

static void join_vals(FILE *f, const double *v, int n) { /* common.c:560-569 */
    for (int i = 0; i < n; i++) fprintf(f, i ? ",%.5e" : "%.5e", v[i]);
}

void ohf_write_emission_tsv(const ohf_model *m, FILE *f) {
    fprintf(f, "#State\tDistribution\tComponents\tParameter");
    for (int r = 0; r < m->n_regions; r++) fprintf(f, "\tValues_Region_%d", r);
    fprintf(f, "\n");
    for (int s = 0; s < OHF_NSTATES; s++) {
        bool te = (s == OHF_STATE_ERR && m->model_type == OHF_MODEL_TRUNC_EXP_GAUSSIAN);
        int np = te ? 2 : 3; /* hmm_utils.c:1539-1575 */
        static const char *TE_NAMES[2] = {"Mean", "Trunc_Point"};
        static const char *G_NAMES[3] = {"Mean", "Var", "Weight"};
        const bool nb = m->model_type == OHF_MODEL_NEGATIVE_BINOMIAL; /* logged as Mean, Var, Weight (hmm_utils.c:1560-1570) */
        for (int p = 0; p < np; p++) {
            fprintf(f, "%s\t%s\t%d\t%s", STATE_NAMES[s], te ? "Truncated Exponential" : nb ? "Negative Binomial" : "Gaussian",
                    te ? 1 : m->ncomp[s], te ? TE_NAMES[p] : G_NAMES[p]);
            for (int r = 0; r < m->n_regions; r++) {
                const ohf_region *g = &m->regions[r];
                fprintf(f, "\t");
                if (te) {
                    double v = p == 0 ? 1.0 / g->lambda : g->trunc_point; /* hmm_utils.c:1056-1069 */
                    join_vals(f, &v, 1);
                } else if (nb && p < 2) { /* hmm_utils.c:589-608, 463-473 */
                    double v[OHF_MAXCOMP];
                    for (int c = 0; c < m->ncomp[s]; c++)
                        v[c] = p == 0 ? ohf_nb_mean(g->theta[s][c], g->nb_lambda[s][c]) : ohf_nb_var(g->theta[s][c], g->nb_lambda[s][c]);
                    join_vals(f, v, m->ncomp[s]);
                } else {
                    join_vals(f, p == 0 ? g->mean[s] : p == 1 ? g->var[s] : g->weight[s], m->ncomp[s]);
                }
            }
            fprintf(f, "\n");
        }
    }
}

/* ---------- final BED (chunk.c:953-1124) ---------- */
static const char *const LABEL_COLORS[] = {"162,0,37", "250,104,0", "0,138,0", "170,0,255", "99, 99, 96", "250,200,0"};
static const char *const LABEL_NAMES[] = {"Err", "Dup", "Hap", "Col", "Unk", "Msj"}; /* chunk.c:10-21 */

typedef struct { int s, e, label; } blk_t;

static void flush_contig(FILE *fout, const char *ctg, blk_t *blocks, int nb) {
    /* chunk.c:953-983 mergeBlocksWithSameLabels (preStart starts at 0) then print */
    int preLabel = -1, preStart = 0, preEnd = 0;
    for (int i = 0; i <= nb; i++) {
        bool last = (i == nb);
        if (!last) {
            if (preLabel != -1 && blocks[i].label != preLabel) {
                fprintf(fout, "%s\t%d\t%d\t%s\t0\t.\t%d\t%d\t%s\n", ctg, preStart, preEnd + 1,
                        LABEL_NAMES[preLabel], preStart, preEnd + 1, LABEL_COLORS[preLabel]);
                preStart = blocks[i].s;
            }
            preEnd = blocks[i].e;
            preLabel = blocks[i].label;
        } else if (nb > 0) {
            fprintf(fout, "%s\t%d\t%d\t%s\t0\t.\t%d\t%d\t%s\n", ctg, preStart, preEnd + 1,
                    LABEL_NAMES[preLabel], preStart, preEnd + 1, LABEL_COLORS[preLabel]);
        }
    }
}

int ohf_write_final_bed(const ohf_chunks *cc, const char *path, const char *track_name,
                        const int min_len_per_state[4]) {
    FILE *fout = fopen(path, "w");
    if (!fout) return -1;
    fprintf(fout, "track name=%s visibility=1 itemRgb=\"On\"\n", track_name);
    const int hapLabel = 2;
    int bedTrackStart = 0, preEnd = 0, preLabel = -1;
    char preCtg[200]; preCtg[0] = '\0';
    blk_t *blocks = NULL; int nb = 0, cap = 0;
    for (int c = 0; c < cc->n_chunks; c++) {
        const ohf_chunk *ch = &cc->chunks[c];
        for (int i = 0; i < ch->n; i++) {
            int start = ch->s + i * cc->window_len;                 /* chunk.c:934-935 */
            int end = ch->s + (i + 1) * cc->window_len - 1;
            if (ch->e < end) end = ch->e;
            if (preLabel == -1 || preCtg[0] == '\0') bedTrackStart = start;
            int predictionLabel = 4;
            if (ch->prediction[i] != -1) predictionLabel = ch->prediction[i];
            bool labelChanged = preLabel != -1 && predictionLabel != preLabel;
            bool contigChanged = preCtg[0] != '\0' && strcmp(preCtg, ch->ctg) != 0;
            if (labelChanged || contigChanged) {
                int blockLen = preEnd + 1 - bedTrackStart;
                if (nb == cap) { cap = cap ? 2 * cap : 1024; blocks = realloc(blocks, sizeof(blk_t) * (size_t) cap); }
                int minLen = preLabel < 4 ? min_len_per_state[preLabel] : 0;
                blocks[nb++] = (blk_t) { bedTrackStart, preEnd, blockLen < minLen ? hapLabel : preLabel };
                bedTrackStart = start;
            }
            if (contigChanged) { flush_contig(fout, preCtg, blocks, nb); nb = 0; }
            preEnd = end;
            preLabel = predictionLabel;
            strcpy(preCtg, ch->ctg);
        }
    }
    if (preLabel != -1) {
        int blockLen = preEnd + 1 - bedTrackStart;
        if (nb == cap) { cap = cap ? 2 * cap : 1024; blocks = realloc(blocks, sizeof(blk_t) * (size_t) cap); }
        int minLen = preLabel < 4 ? min_len_per_state[preLabel] : 0;
        blocks[nb++] = (blk_t) { bedTrackStart, preEnd, blockLen < minLen ? hapLabel : preLabel };
        flush_contig(fout, preCtg, blocks, nb);
    }
    free(blocks);
    fclose(fout);
    return 0;
}

/* ---------- posterior BED (src/hmm_flagger.c:240-282) ---------- */
int ohf_write_posterior_bed(const ohf_chunks *cc, const char *path) {
    FILE *fout = fopen(path, "w");
    if (!fout) return -1;
    fprintf(fout, "#ctg\tstart\tend\t");
    for (int s = 0; s < OHF_NSTATES; s++) fprintf(fout, "posterior_%s_%d\t", STATE_NAMES[s], s);
    fprintf(fout, "prediction\n");
    for (int c = 0; c < cc->n_chunks; c++) {
        const ohf_chunk *ch = &cc->chunks[c];
        for (int i = 0; i < ch->n; i++) {
            int start = ch->s + i * cc->window_len;
            int end = ch->s + (i + 1) * cc->window_len - 1;
            if (ch->e < end) end = ch->e;
            fprintf(fout, "%s\t%d\t%d\t", ch->ctg, start, end + 1);
            double post[4];
            ohf_posterior(ch, i, post);
            int pred = ohf_most_probable_state(ch, i);
            for (int s = 0; s < OHF_NSTATES; s++) fprintf(fout, "%.2f\t", post[s]);
            fprintf(fout, "%s\n", STATE_NAMES[pred]);
        }
    }
    fclose(fout);
    return 0;
}
