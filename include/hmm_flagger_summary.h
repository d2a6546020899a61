/*
 * hmm_flagger_summary.h — C ABI of the prediction summary tables, the third output of the
 * drop-in next to the BED and the parameter TSVs (wdls/tasks/hmm_flagger/hmm_flagger.wdl:102).
 *
 * Reference interface replaced (mobinasri/flagger, programs/):
 *   writeBenchmarkingStats                          src/hmm_flagger.c:134-162
 *   SummaryTableList_createAndWriteAllTables        submodules/summary_table/summary_table.c:1663-1747
 *     -> SummaryTableList_updateByUpdaterArgs       summary_table.c:934-1223   (one pass per category index)
 *     -> SummaryTableListFullCatalog_write          summary_table.c:1403-1586
 *   IntBinArray_constructFromFile / _getBinIndices  submodules/common/common.c:688-751
 * Host-side, label-run based; no device work.
 */
#ifndef HMM_FLAGGER_SUMMARY_H
#define HMM_FLAGGER_SUMMARY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The windows in chunk-list order, exactly what ChunkIterator_getNextPtBlock walks (chunk.c:915-950):
 * window i of chunk c covers [s + i*W, min(s + (i+1)*W - 1, e)] on contig chunk_ctg[c]. */
typedef struct hfs_input {
    int64_t n_windows;
    int32_t n_chunks;
    const int64_t *chunk_off;             /* [n_chunks + 1] */
    const int32_t *chunk_s, *chunk_e;     /* [n_chunks] 0-based inclusive */
    const char *const *chunk_ctg;         /* [n_chunks] */
    int32_t window_len;
    const uint64_t *annot;                /* [n_windows] annotation bits + region index in bits 58..63 */
    const int8_t *truth;                  /* [n_windows] or NULL; -1 = unknown */
    const int8_t *prediction;             /* [n_windows] or NULL; -1 = unknown */
    int32_t truth_available, prediction_available;   /* CoverageHeader.isTruthAvailable / isPredictionAvailable */
    int32_t n_labels;                     /* CoverageHeader.numberOfLabels (4 once the HMM has run, hmm_flagger.c:354) */
    int32_t n_regions;                    /* category names region_0 .. (track_reader.c:83-93) */
    int32_t n_annotations;
    const char *const *annotation_names;  /* [n_annotations] */
} hfs_input;

/* Writes <output_path> (must end in ".tsv") and, when both truth and prediction are available,
 * <prefix>.benchmarking.tsv and <prefix>.benchmarking.auN_ratio.tsv.
 *   bin_array_path            --binArrayFile: TSV "start<TAB>end<TAB>name" (NULL: one bin [0, 1e9) "ALL_SIZES")
 *   label_names_with_unknown  --labelNames + "Unk" (NULL / 0: rows are numbered, the header line has no label columns —
 *                             the reference's behaviour, summary_table.c:1419-1434)
 *   overlap_ratio_threshold   --overlapRatioThreshold (default 0.4, hmm_flagger.c:632)
 * Returns 0, or -1 with a message in hfs_last_error(). */
int hfs_write_all_tables(const hfs_input *in, const char *output_path, const char *bin_array_path,
                         const char *const *label_names_with_unknown, int n_label_names,
                         double overlap_ratio_threshold, int threads);
const char *hfs_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
