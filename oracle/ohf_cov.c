/*
 * ohf_cov.c — ORACLE (test infrastructure, not product code).
 * `.cov` / `.cov.gz` reader + chunking + windowing, restated from the reference with its per-base
 * accumulation (citations: file:line under /root/reference/programs/submodules/).
 *   header lines            track_reader/track_reader.c:48-457
 *   rows                    track_reader/track_reader.c:751-818  (1-based inclusive -> 0-based)
 *   chunk index             chunk/chunk.c:240-294
 *   windows                 chunk/chunk.c:393-483, 506-547
 */
#include "ohf.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define LINE_MAX_SIZE 8192
#define MAX_COVERAGE 250 /* chunk.c:8 */

typedef struct {
    char ctg[200];
    int ctg_len, s, e; /* 0-based inclusive */
    double cov, mapq, clip;
    uint64_t annot_flag;
    int region, truth, prediction;
} row_t;

/* common.c:407-427 Int_getModeValue1DArray: lowest value wins ties */
static int mode_value(const int *a, int len, int minv, int maxv) {
    int n = maxv - minv + 1;
    int *counts = calloc((size_t) n, sizeof(int));
    for (int i = 0; i < len; i++) {
        int idx = a[i] < minv ? 0 : a[i] - minv;
        idx = n <= idx ? n - 1 : idx;
        counts[idx] += 1;
    }
    int mode = minv, maxc = counts[0];
    for (int i = 1; i < n; i++)
        if (maxc < counts[i]) { mode = minv + i; maxc = counts[i]; }
    free(counts);
    return mode;
}

static uint64_t annot_flag_from_list(const char *s) { /* ptBlock.c:225-236 */
    uint64_t flag = 0;
    char buf[1024];
    strncpy(buf, s, sizeof(buf) - 1);
    buf[sizeof(buf) - 1] = '\0';
    char *save = NULL;
    for (char *tok = strtok_r(buf, ",", &save); tok; tok = strtok_r(NULL, ",", &save)) {
        int idx = atoi(tok);
        if (0 < idx) flag |= 1ULL << (idx - 1);
    }
    return flag;
}

typedef struct { row_t *rows; int n, cap; } rows_t;

static int parse_file(const char *path, ohf_chunks *cc, rows_t *out) {
    gzFile f = gzopen(path, "r"); /* transparently reads plain text too */
    if (!f) return -1;
    char *line = malloc(LINE_MAX_SIZE);
    char ctg[200] = "";
    int ctg_len = 0;
    int n_parsed_cov = 0;
    bool have_ann_len = false, have_reg_len = false;
    cc->n_annotations = 0; cc->n_regions = 0; cc->n_labels = 0; cc->avg_alignment_len = 0;
    while (gzgets(f, line, LINE_MAX_SIZE)) {
        size_t L = strlen(line);
        if (L > 0 && line[L - 1] == '\n') line[--L] = '\0';
        if (L == 0) continue;
        if (line[0] == '#') {
            if (!strncmp(line, "#annotation:len", 15) && !have_ann_len) {
                const char *p = strchr(line + 12, ':');
                cc->n_annotations = p ? atoi(p + 1) : 0; have_ann_len = true;
                cc->annotation_names = calloc((size_t) (cc->n_annotations > 0 ? cc->n_annotations : 1) + 1, sizeof(char *));
                for (int i = 0; i < cc->n_annotations; i++) cc->annotation_names[i] = strdup("NA");
            } else if (!strncmp(line, "#annotation:name:", 17) && cc->annotation_names) {
                int idx = atoi(line + 17);
                const char *p = strchr(line + 17, ':');
                if (p && idx >= 0 && idx < cc->n_annotations) {
                    char nm[512]; strncpy(nm, p + 1, sizeof(nm) - 1); nm[sizeof(nm) - 1] = '\0';
                    char *q = strchr(nm, ':'); if (q) *q = '\0';
                    free(cc->annotation_names[idx]); cc->annotation_names[idx] = strdup(nm);
                }
            } else if (!strncmp(line, "#region:len", 11) && !have_reg_len) {
                const char *p = strchr(line + 8, ':');
                cc->n_regions = p ? atoi(p + 1) : 0; have_reg_len = true;
            } else if (!strncmp(line, "#region:coverage:", 17)) {
                int idx = atoi(line + 17);
                const char *p = strchr(line + 17, ':');
                if (p && idx >= 0 && idx < OHF_MAXREGIONS) { cc->region_coverages[idx] = atoi(p + 1); n_parsed_cov++; }
            } else if (!strncmp(line, "#label:len", 10)) {
                const char *p = strchr(line + 7, ':');
                if (p && cc->n_labels == 0) cc->n_labels = atoi(p + 1);
            } else if (!strncmp(line, "#truth:true", 11)) cc->truth_available = true;
            else if (!strncmp(line, "#prediction:true", 16)) cc->prediction_available = true;
            else if (!strncmp(line, "#start-only:true", 16)) cc->start_only = true;
            else if (!strncmp(line, "#avg_alignment_len:", 19) && cc->avg_alignment_len == 0) cc->avg_alignment_len = atoi(line + 19);
            continue;
        }
        if (line[0] == '>') {
            char *sp = strchr(line, ' ');
            if (sp) { *sp = '\0'; ctg_len = atoi(sp + 1); } else ctg_len = 0;
            strncpy(ctg, line + 1, sizeof(ctg) - 1);
            continue;
        }
        /* start end cov mapq clip annots region [truth [prediction]] */
        char *fld[16]; int nf = 0; char *save = NULL;
        for (char *tok = strtok_r(line, "\t", &save); tok && nf < 16; tok = strtok_r(NULL, "\t", &save)) fld[nf++] = tok;
        if (nf < 7) { free(line); gzclose(f); return -2; }
        if (out->n == out->cap) { out->cap = out->cap ? 2 * out->cap : 4096; out->rows = realloc(out->rows, sizeof(row_t) * (size_t) out->cap); }
        row_t *r = &out->rows[out->n++];
        strcpy(r->ctg, ctg); r->ctg_len = ctg_len;
        r->s = atoi(fld[0]) - 1; r->e = atoi(fld[1]) - 1;
        r->cov = atof(fld[2]); r->mapq = atof(fld[3]); r->clip = atof(fld[4]);
        r->annot_flag = annot_flag_from_list(fld[5]);
        r->region = atoi(fld[6]);
        r->truth = nf >= 8 ? atoi(fld[7]) : -1;       /* chunk.c:471-474: attrbsLen counts fields after start/end */
        r->prediction = nf >= 9 ? atoi(fld[8]) : -1;
    }
    free(line);
    gzclose(f);
    if (!have_ann_len || cc->n_annotations <= 0 || !have_reg_len || cc->n_regions <= 0 || n_parsed_cov != cc->n_regions) return -3;
    return 0;
}

typedef struct {
    int window_len, itr; /* itr = windowItr */
    double sum_cov, sum_mapq, sum_clip;
    uint64_t flag;
    int *reg, *tru, *pre;
    bool start_only;
} win_t;

static void push_window(ohf_chunk *ch, win_t *w, int *cap) { /* chunk.c:393-441 */
    if (w->itr == -1) return;
    double c, m, k;
    int n = w->itr + 1;
    if (w->start_only) {
        c = (double) w->sum_cov * w->window_len / n;
        m = (double) w->sum_mapq * w->window_len / n;
        k = (double) w->sum_clip * w->window_len / n;
    } else {
        c = (double) w->sum_cov / n; m = (double) w->sum_mapq / n; k = (double) w->sum_clip / n;
    }
    if (ch->n == *cap) {
        *cap = *cap ? 2 * *cap : 256;
        ch->cov = realloc(ch->cov, 2 * (size_t) *cap); ch->mapq = realloc(ch->mapq, 2 * (size_t) *cap);
        ch->clip = realloc(ch->clip, 2 * (size_t) *cap); ch->annot = realloc(ch->annot, 8 * (size_t) *cap);
        ch->truth = realloc(ch->truth, (size_t) *cap); ch->prediction = realloc(ch->prediction, (size_t) *cap);
    }
    int i = ch->n++;
    ch->cov[i] = MAX_COVERAGE < round(c) ? MAX_COVERAGE : round(c);
    ch->mapq[i] = MAX_COVERAGE < round(m) ? MAX_COVERAGE : round(m);
    ch->clip[i] = MAX_COVERAGE < round(k) ? MAX_COVERAGE : round(k);
    int8_t truth = (int8_t) mode_value(w->tru, n, -1, 10);       /* chunk.c:377-391 */
    int8_t pred = (int8_t) mode_value(w->pre, n, -1, 10);
    int region = mode_value(w->reg, n, 0, 100);
    uint64_t a = w->flag & 0x03FFFFFFFFFFFFFFULL;                 /* ptBlock.c:300-304 */
    a |= ((uint64_t) region) << 58;
    ch->annot[i] = a; ch->truth[i] = truth; ch->prediction[i] = pred;
    w->itr = -1; w->flag = 0; w->sum_cov = w->sum_mapq = w->sum_clip = 0.0;
}

ohf_chunks *ohf_read_cov(const char *path, int chunk_len, int window_len) {
    ohf_chunks *cc = calloc(1, sizeof(ohf_chunks));
    rows_t rows = {0};
    if (parse_file(path, cc, &rows) != 0) { free(rows.rows); ohf_chunks_destroy(cc); return NULL; }
    cc->chunk_len = chunk_len; cc->window_len = window_len;
    int cap_chunks = 0;
    win_t w = { window_len, -1, 0, 0, 0, 0, malloc(sizeof(int) * (size_t) window_len),
                malloc(sizeof(int) * (size_t) window_len), malloc(sizeof(int) * (size_t) window_len), cc->start_only };
    int i = 0;
    while (i < rows.n) {
        /* rows of one contig: [i, j) */
        int j = i;
        while (j < rows.n && !strcmp(rows.rows[j].ctg, rows.rows[i].ctg)) j++;
        int ctg_len = rows.rows[i].ctg_len;
        /* chunk bounds, chunk.c:259-286 */
        int s = 0, e = ctg_len < 2 * chunk_len ? ctg_len - 1 : chunk_len - 1;
        int r = i;
        for (;;) {
            if (cc->n_chunks == cap_chunks) { cap_chunks = cap_chunks ? 2 * cap_chunks : 64; cc->chunks = realloc(cc->chunks, sizeof(ohf_chunk) * (size_t) cap_chunks); }
            ohf_chunk *ch = &cc->chunks[cc->n_chunks++];
            memset(ch, 0, sizeof(*ch));
            strcpy(ch->ctg, rows.rows[i].ctg); ch->ctg_len = ctg_len; ch->s = s; ch->e = e;
            int cap = 0;
            w.itr = -1; w.flag = 0; w.sum_cov = w.sum_mapq = w.sum_clip = 0.0;
            /* chunk.c:534-545 + 444-483: add every base of every overlapping row */
            for (int q = r; q < j; q++) {
                const row_t *t = &rows.rows[q];
                if (ch->s <= t->e && t->s <= ch->e) {
                    int b0 = t->s > ch->s ? t->s : ch->s, b1 = t->e < ch->e ? t->e : ch->e;
                    for (int b = b0; b <= b1; b++) {
                        w.itr += 1; w.itr %= window_len;
                        w.sum_cov += t->cov; w.sum_mapq += t->mapq; w.sum_clip += t->clip;
                        w.flag |= t->annot_flag;
                        w.reg[w.itr] = t->region; w.tru[w.itr] = t->truth; w.pre[w.itr] = t->prediction;
                        if (w.itr == window_len - 1) push_window(ch, &w, &cap);
                    }
                }
                if (ch->e <= t->e) { if (w.itr != -1) push_window(ch, &w, &cap); r = q; break; }
            }
            if (e >= ctg_len - 1) break;
            s = e + 1;
            e = ctg_len < (s - 1) + 2 * chunk_len ? ctg_len - 1 : (s - 1) + chunk_len;
        }
        i = j;
    }
    free(w.reg); free(w.tru); free(w.pre); free(rows.rows);
    return cc;
}
