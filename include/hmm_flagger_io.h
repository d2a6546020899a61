/*
 * hmm_flagger_io.h — C ABI of the host-side data formats either side of the hot path
 * (SURVEY.md §8f N1 and Appendix B): the window table the E-step consumes and the BED it feeds.
 *
 * Reference interfaces replaced (mobinasri/flagger, programs/submodules/):
 *   ChunksCreator_constructFromCov + ChunksCreator_parseChunks   chunk/chunk.c:141-202, 486-547
 *   CoverageHeader_construct                                     track_reader/track_reader.c:48-81
 *   ChunksCreator_parseChunksFromBinaryFile / writeChunksIntoBinaryFile   chunk/chunk.c:596-828
 *   ChunksCreator_writePredictionIntoFinalBED                    chunk/chunk.c:985-1124
 *   writePosteriorIntoBED                                        src/hmm_flagger.c:240-282
 * The loader is run-length aware: O(rows) instead of the reference's O(bases) with several atof per base.
 */
#ifndef HMM_FLAGGER_IO_H
#define HMM_FLAGGER_IO_H

#include <stdint.h>
#include <stddef.h>
#include "hmm_flagger_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hfio_table hfio_table;

/* `path` ends in .cov, .cov.gz or .bin (common.c:51-66 extension rule); chunk_len/window_len are ignored
 * for .bin (the file's own values win, hmm_flagger.c:82-90).  NULL on error (hfio_last_error()). */
hfio_table *hfio_load(const char *path, int chunk_len, int window_len);
void hfio_destroy(hfio_table *t);
const char *hfio_last_error(void);

int64_t hfio_n_windows(const hfio_table *t);
int32_t hfio_n_chunks(const hfio_table *t);
int32_t hfio_n_regions(const hfio_table *t);
const int32_t *hfio_region_coverages(const hfio_table *t);
int32_t hfio_window_len(const hfio_table *t);
int32_t hfio_chunk_len(const hfio_table *t);
int32_t hfio_avg_alignment_len(const hfio_table *t);
int32_t hfio_start_only(const hfio_table *t);
int32_t hfio_n_annotations(const hfio_table *t);
const char *hfio_annotation_name(const hfio_table *t, int i);
const char *hfio_chunk_ctg(const hfio_table *t, int c);
/* --contigsList: keep only the chunks whose contig is named (ChunksCreator_subsetChunksToContigs, chunk.c:218-237);
 * chunk order is preserved.  Returns the number of chunks kept. */
int32_t hfio_subset_contigs(hfio_table *t, const char *const *names, int n_names);
/* first space-delimited token of every line (Splitter_parseLinesIntoList, common.c:620-642); NULL-terminated
 * array owned by the caller (free each entry and the array). */
char **hfio_read_name_list(const char *path, int *n_names);
int8_t *hfio_truth(hfio_table *t);          /* [n_windows] */
int8_t *hfio_prediction(hfio_table *t);     /* [n_windows], -1 until set */

/* Fill the pointer fields and window_len/mean_read_len of `w` from the table (arrays stay owned by `t`);
 * the caller sets the run options (adjust_contig_ends, ratios). */
void hfio_windows(const hfio_table *t, hf_windows *w);

int hfio_write_bin(const hfio_table *t, const char *path);
/* labels: [n_windows] state indices (-1 = unknown); min_len_per_state[4] = --minimumLengths mapped to
 * states Err,Dup,Hap,Col (hmm_flagger.c:737-748). */
int hfio_write_final_bed(const hfio_table *t, const int8_t *labels, const char *path, const char *track_name,
                         const int32_t *min_len_per_state);
/* posterior: [n_windows][4] */
/* prediction_summary_<suffix>.tsv (+ .benchmarking.tsv / .benchmarking.auN_ratio.tsv when the truth track is
 * present) for the table's windows with `labels` as the prediction: writeBenchmarkingStats, hmm_flagger.c:134-162.
 * Marks the prediction as available with 4 labels, as hmm_flagger.c:353-354 does.  See hmm_flagger_summary.h. */
int hfio_write_summary(hfio_table *t, const int8_t *labels, const char *output_path, const char *bin_array_path,
                       const char *const *label_names_with_unknown, int n_label_names, double overlap_ratio_threshold,
                       int threads);
int32_t hfio_truth_available(const hfio_table *t);
int32_t hfio_n_labels(const hfio_table *t);
int hfio_write_posterior_bed(const hfio_table *t, const double *posterior, const int8_t *labels, const char *path);
/* The loader's own DEFLATE / gzip decoder (csrc/hf_inflate.h) on a whole file, for tests and tools: every member of `path`
 * decoded into one malloc'ed buffer (*out, *n; free with hfio_free), each member's CRC-32 and ISIZE verified.
 * 0, or -1 data error / -2 truncated / -3 not a gzip file / -4 CRC-32 or length mismatch / -5 cannot read the file. */
int hfio_gunzip(const char *path, unsigned char **out, size_t *n);
void hfio_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
