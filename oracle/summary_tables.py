"""oracle/summary_tables.py — TEST INFRASTRUCTURE ONLY (checker for flagger_amd/csrc/hf_summary.cpp; never imported
by the product).  Pinned by the reference's own known answers: tests/test_reference_kats_cpu.py reproduces the expected
tables / strings of programs/tests/test_summary_table.c:12-470 and test_common.c:84-127 on the reference's data files with this
module (and with the product).  The final-statistics files (benchmarking.tsv, auN) have no reference-held vector.

Literal pure-Python restatement (small inputs only) of the reference's prediction summary tables,
mobinasri/flagger programs/submodules/summary_table/summary_table.c:
  SummaryTable_increment                          :70-92    (percentages refreshed at every increment)
  convertBaseLevelToOverlapBased                  :817-834
  SummaryTableList_updateByUpdaterArgs            :934-1223 (one sequential pass per category index)
  SummaryTableList_addCreationJobsForOneMetricType / _createAndWriteAllTables   :1588-1747
  SummaryTableListFullCatalog_write               :1403-1586
  SummaryTableList_writeFinalStatisticsIntoFile   :457-741, _writeFinalAunStatisticsIntoFile :744-811
and of the iterator it walks (ChunkIterator_getNextPtBlock, chunk/chunk.c:915-950), the category tests
(ptBlock/ptBlock.c:225-255, 282-298) and the size bins (common/common.c:620-751).
"""
import os

OVERLAP_BASED, BASE_LEVEL, AUN = 0, 1, 2
CAT_REGION, CAT_ANNOTATION = 0, 1
TRUTH_VS_PRED, PRED_VS_TRUTH, TRUTH_VS_TRUTH, PRED_VS_PRED = 0, 1, 2, 3
METRIC_NAME = ["overlap_based", "base_level", "truth_based_auN"]
CATEGORY_NAME = ["region", "annotation"]
COMPARISON_NAME = ["TRUTH_VS_PREDICTION", "PREDICTION_VS_TRUTH", "TRUTH", "PREDICTION"]
HAP_INDEX = 2
REGION_MASK = 0xFC00000000000000


class SummaryTable:
    def __init__(self, nr, nc):
        self.nr, self.nc = nr, nc
        self.table = [[0.0] * nc for _ in range(nr)]
        self.pct = [[0.0] * nc for _ in range(nr)]
        self.row_total = [0.0] * nr
        self.row_total_pct = [0.0] * nr
        self.total = 0.0

    def increment(self, r, c, v):                       # summary_table.c:70-92
        self.table[r][c] += v
        self.row_total[r] += v
        self.total += v
        for j in range(self.nc):
            if 0 < self.row_total[r]:
                self.pct[r][j] = self.table[r][j] / self.row_total[r] * 100.0
        if 0 < self.total:
            for i in range(self.nr):
                self.row_total_pct[i] = self.row_total[i] / self.total * 100.0


class TableList:
    def __init__(self, names1, names2, nrows):
        self.names1, self.names2, self.nrows = names1, names2, nrows
        self.tabs = [[SummaryTable(nrows, nrows) for _ in names2] for _ in names1]


def read_bins(path):
    if path is None:
        return [0], [int(1e9)], ["ALL_SIZES"]
    starts, ends, names = [], [], []
    for line in open(path):
        line = line.rstrip("\n").split(" ")[0]          # Splitter_parseLinesIntoList keeps the first space-delimited token
        if not line or line[0] == "#":
            continue
        tok = line.split("\t")
        starts.append(int(float(tok[0])))
        ends.append(int(float(tok[1])))
        names.append(tok[2])
    return starts, ends, names


def overlaps(flag, cat_type, c1):
    if flag is None:
        return False
    if cat_type == CAT_REGION:                          # ptBlock.c:252-255
        return c1 == ((flag & REGION_MASK) >> 58)
    if (flag & ~REGION_MASK & 0xFFFFFFFFFFFFFFFF) == 0 and c1 == 0:   # ptBlock.c:245-250
        return True
    bit = (1 << (c1 - 1)) if 0 < c1 else 0
    return (bit & flag) != 0


def iterate_windows(inp):
    """ChunkIterator_getNextPtBlock: (ctg, start, end, annotation flag, truth, prediction) per window."""
    W = inp["window_len"]
    for c in range(len(inp["chunk_ctg"])):
        t0, t1 = inp["chunk_off"][c], inp["chunk_off"][c + 1]
        s, e = inp["chunk_s"][c], inp["chunk_e"][c]
        for i in range(t1 - t0):
            t = t0 + i
            yield (inp["chunk_ctg"][c], s + i * W, min(s + (i + 1) * W - 1, e), int(inp["annot"][t]),
                   int(inp["truth"][t]) if inp["truth"] is not None else -1,
                   int(inp["prediction"][t]) if inp["prediction"] is not None else -1)


def update(inp, tl, aux, bins, cat_type, c1, metric, cmp_, thr):
    """SummaryTableList_updateByUpdaterArgs, summary_table.c:934-1223"""
    starts, ends, _ = bins
    nc = tl.nrows
    row = [0.0] * nc
    qlens = [[] for _ in range(nc)]
    ref_start = qry_start = pre_ref = pre_qry = pre_end = -1
    pre_flag, pre_ctg = None, ""
    ref_is_truth = cmp_ in (TRUTH_VS_PRED, TRUTH_VS_TRUTH)
    qry_is_pred = cmp_ in (TRUTH_VS_PRED, PRED_VS_PRED)

    def flush():
        block_len = pre_end - ref_start + 1
        idx = [i for i in range(len(starts)) if starts[i] <= block_len < ends[i]]
        if metric == OVERLAP_BASED:
            hit = False
            for i in range(nc):
                ratio = row[i] / block_len
                if thr < ratio:
                    hit = True
                row[i] = 1 if thr < ratio else 0
            if not hit:
                row[nc - 1] = 1
        if metric == AUN:
            if pre_qry != -1:
                qlens[pre_qry].append(pre_end - qry_start + 1)
            for q in range(nc):
                for ln in qlens[q]:
                    row[q] += float(ln) * ln
        for b in idx:
            total = aux.tabs[c1][b].table[pre_ref][pre_ref] if metric == AUN else 1.0
            for q in range(nc):
                tl.tabs[c1][b].increment(pre_ref, q, row[q] / total)

    for ctg, start, end, flag, truth, pred in iterate_windows(inp):
        ref = truth if ref_is_truth else pred
        qry = pred if qry_is_pred else truth
        if ref == -1:
            ref = tl.nrows - 1
        if qry == -1:
            qry = nc - 1
        contig_changed = pre_ctg != "" and pre_ctg != ctg
        ref_changed, qry_changed = ref != pre_ref, qry != pre_qry
        in_cur, in_prev = overlaps(flag, cat_type, c1), overlaps(pre_flag, cat_type, c1)
        continued, started, ended = in_cur and in_prev, in_cur and not in_prev, (not in_cur) and in_prev
        pre_ref_valid, pre_qry_valid = pre_ref != -1, pre_qry != -1
        if pre_ref_valid and ((continued and ref_changed) or (in_prev and contig_changed) or ended):
            flush()
        if in_cur and metric == AUN and pre_qry_valid and qry_changed and (continued and not ref_changed) and not contig_changed:
            qlens[pre_qry].append(pre_end - qry_start + 1)
        if ((not in_cur) and contig_changed) or ended:
            ref_start = qry_start = -1
            row[:] = [0.0] * nc
        if (continued and ref_changed) or (in_cur and contig_changed) or started:
            ref_start = start
            row[:] = [0.0] * nc
            for q in range(nc):
                qlens[q] = []
        if (continued and ref_changed) or (continued and qry_changed) or (in_cur and contig_changed) or started:
            qry_start = start
        if in_cur and metric != AUN:
            row[qry] += end - start + 1
        pre_flag, pre_ref, pre_qry, pre_ctg, pre_end = flag, ref, qry, ctg, end
    if overlaps(pre_flag, cat_type, c1) and pre_ref != -1:
        flush()


def _join(vals):
    return "\t".join("%.2f" % v for v in vals)


def _na(ok, v):
    return "%.2f" % v if ok else "NA"


def write_all_tables(inp, output_path, bin_array_path=None, label_names_with_unknown=None, overlap_ratio_threshold=0.4):
    """SummaryTableList_createAndWriteAllTables + SummaryTableListFullCatalog_write"""
    bins = read_bins(bin_array_path)
    n_rows = inp["n_labels"] + 1
    names = {CAT_REGION: ["region_%d" % i for i in range(inp["n_regions"])], CAT_ANNOTATION: list(inp["annotation_names"])}
    truth, pred = bool(inp["truth_available"]), bool(inp["prediction_available"])
    cat = {}

    def add_jobs(metric):
        for ct in (CAT_REGION, CAT_ANNOTATION):
            for cmp_ in range(4):
                need_t = cmp_ in (TRUTH_VS_PRED, PRED_VS_TRUTH, TRUTH_VS_TRUTH)
                need_p = cmp_ in (TRUTH_VS_PRED, PRED_VS_TRUTH, PRED_VS_PRED)
                if (not truth and need_t) or (not pred and need_p):
                    continue
                if metric == AUN and cmp_ in (PRED_VS_PRED, PRED_VS_TRUTH):
                    continue
                tl = TableList(names[ct], bins[2], n_rows)
                cat[(ct, metric, cmp_)] = tl
                aux = cat.get((ct, BASE_LEVEL, TRUTH_VS_TRUTH)) if metric == AUN else None
                for c1 in range(len(names[ct])):
                    update(inp, tl, aux, bins, ct, c1, metric, cmp_, overlap_ratio_threshold)

    if truth or pred:
        add_jobs(OVERLAP_BASED)
        add_jobs(BASE_LEVEL)
        add_jobs(AUN)

    labels = label_names_with_unknown
    rowname = (lambda r: labels[r]) if labels else (lambda r: "%d" % r)
    fout = open(output_path, "w")
    fout.write("#Statistic\tMetric_Type\tEntry_Type\tCategory_Type\tCategory_Name\tSize_Bin_Name\tRef_Label")
    for l in labels or []:
        fout.write("\t" + l)
    fout.write("\n")
    fstats = faun = None
    if truth and pred:
        prefix = output_path[:-4]
        fstats = open(prefix + ".benchmarking.tsv", "w")
        fstats.write("#Metric_Type\tCategory_Type\tCategory_Name\tSize_Bin_Name\tLabel\tTP_Prediction_Ref\tTP_Truth_Ref\tFP\tFN\t"
                     "Total_Prediction_Ref\tTotal_Truth_Ref\tPrecision\tRecall\tF1-Score\tAccuracy_Prediction_Ref\tAccuracy_Truth_Ref\n")
        faun = open(prefix + ".benchmarking.auN_ratio.tsv", "w")
        faun.write("#Category_Type\tCategory_Name\tSize_Bin_Name\tLabel\tauN_Ratio\n")
    for ct in (CAT_REGION, CAT_ANNOTATION):
        for metric in range(3):
            for cmp_ in range(4):
                tl = cat.get((ct, metric, cmp_))
                if tl is None:
                    continue
                total_only = cmp_ in (TRUTH_VS_TRUTH, PRED_VS_PRED)
                for entry in ("count", "percentage"):
                    prefix = "%s\t%s\t%s\t%s" % (COMPARISON_NAME[cmp_], METRIC_NAME[metric], entry, CATEGORY_NAME[ct])
                    for c1, n1 in enumerate(tl.names1):
                        for c2, n2 in enumerate(tl.names2):
                            t = tl.tabs[c1][c2]
                            if total_only:
                                vals = t.row_total if entry == "count" else t.row_total_pct
                                fout.write("%s\t%s\t%s\t%s\n" % (prefix, n1, n2, "ALL_LABELS\t" + _join(vals)))
                            else:
                                for r in range(t.nr):
                                    vals = t.table[r] if entry == "count" else t.pct[r]
                                    fout.write("%s\t%s\t%s\t%s\n" % (prefix, n1, n2, rowname(r) + "\t" + _join(vals)))
            if truth and pred and metric != AUN:
                _final_stats(fstats, cat[(ct, metric, TRUTH_VS_PRED)], cat[(ct, metric, PRED_VS_TRUTH)], rowname,
                             "%s\t%s" % (METRIC_NAME[metric], CATEGORY_NAME[ct]))
        if truth and pred:
            _final_aun(faun, cat[(ct, AUN, TRUTH_VS_PRED)], cat[(ct, AUN, TRUTH_VS_TRUTH)], rowname, CATEGORY_NAME[ct])
    fout.close()
    if fstats:
        fstats.close()
        faun.close()


def _final_stats(f, recall, precision, rowname, prefix):     # summary_table.c:457-741
    n_labels = recall.nrows - 1
    for c1, n1 in enumerate(recall.names1):
        for c2, n2 in enumerate(recall.names2):
            rt, pt = recall.tabs[c1][c2], precision.tabs[c1][c2]
            tot_tp_r = tot_tp_p = tot_r = tot_p = 0.0
            sum_r = sum_p = sum_r_nh = sum_p_nh = 0.0
            rec_r = rec_p = rec_r_nh = rec_p_nh = 0.0
            nz_r = nz_p = nz_r_nh = nz_p_nh = 0
            for r in range(n_labels):
                tp_r, tp_p = rt.table[r][r], pt.table[r][r]
                tot_tp_r += tp_r
                tot_tp_p += tp_p
                fn, fp = rt.row_total[r] - tp_r, pt.row_total[r] - tp_p
                tot_r += tp_r + fn
                tot_p += tp_p + fp
                rec = tp_r / (tp_r + fn + 1.0e-9) * 100.0
                pre = tp_p / (tp_p + fp + 1.0e-9) * 100.0
                r_ok, p_ok = 1e-9 < (tp_r + fn), 1e-9 < (tp_p + fp)
                nz_r += r_ok
                nz_p += p_ok
                if r != HAP_INDEX:
                    nz_r_nh += r_ok
                    nz_p_nh += p_ok
                sum_r += rec
                sum_p += pre
                if r != HAP_INDEX:
                    sum_r_nh += rec
                    sum_p_nh += pre
                if r_ok:
                    v = 1.0 / rec if 0.0 < rec else 1.0e9
                    rec_r += v
                    if r != HAP_INDEX:
                        rec_r_nh += v
                if p_ok:
                    v = 1.0 / pre if 0.0 < pre else 1.0e9
                    rec_p += v
                    if r != HAP_INDEX:
                        rec_p_nh += v
                f1 = 2 * pre * rec / (pre + rec + 1.0e-9)
                f.write("%s\t%s\t%s\t%s\t%.2f\t%.2f\t%.2f\t%.2f\t%.2f\t%.2f\t%s\t%s\t%s\t%s\t%s\n" % (
                    prefix, n1, n2, rowname(r), tp_p, tp_r, fp, fn, tp_p + fp, tp_r + fn, _na(p_ok, pre), _na(r_ok, rec),
                    _na(r_ok and p_ok, f1), "NA", "NA"))
            mac_r = sum_r / nz_r if 0 < nz_r else 0.0
            mac_p = sum_p / nz_p if 0 < nz_p else 0.0
            mac_r_nh = sum_r_nh / nz_r_nh if 0 < nz_r_nh else 0.0
            mac_p_nh = sum_p_nh / nz_p_nh if 0 < nz_p_nh else 0.0
            har_r = float(nz_r) / rec_r if 0 < nz_r else 0.0
            har_p = float(nz_p) / rec_p if 0 < nz_p else 0.0
            har_r_nh = float(nz_r_nh) / rec_r_nh if 0 < nz_r_nh else 0.0
            har_p_nh = float(nz_p_nh) / rec_p_nh if 0 < nz_p_nh else 0.0

            def line(name, p_ok, p, r_ok, r):
                f1 = 2 * r * p / (r + p + 1.0e-9)
                f.write("%s\t%s\t%s\t%s\tNA\tNA\tNA\tNA\tNA\tNA\t%s\t%s\t%s\tNA\tNA\n" % (
                    prefix, n1, n2, name, _na(p_ok, p), _na(r_ok, r), _na(p_ok and r_ok, f1)))
            line("MACRO_AVERAGE", 0 < nz_p, mac_p, 0 < nz_r, mac_r)
            line("MACRO_AVERAGE_NO_HAP", 0 < nz_p_nh, mac_p_nh, 0 < nz_r_nh, mac_r_nh)
            line("HARMONIC_MEAN", 0 < nz_p, har_p, 0 < nz_r, har_r)
            line("HARMONIC_MEAN_NO_HAP", 0 < nz_p_nh, har_p_nh, 0 < nz_r_nh, har_r_nh)
            acc_p = tot_tp_p / (tot_p + 1.0e-9) * 100.0
            acc_r = tot_tp_r / (tot_r + 1e-9) * 100.0
            f.write("%s\t%s\t%s\tACCURACY\t%.2f\t%.2f\tNA\tNA\t%.2f\t%.2f\tNA\tNA\tNA\t%.2f\t%.2f\n" % (
                prefix, n1, n2, tot_tp_p, tot_tp_r, tot_p, tot_r, acc_p, acc_r))


def _final_aun(f, num, den, rowname, prefix):               # summary_table.c:744-811
    n_labels = num.nrows - 1
    for c1, n1 in enumerate(num.names1):
        for c2, n2 in enumerate(num.names2):
            nt, dt = num.tabs[c1][c2], den.tabs[c1][c2]
            s = rec = 0.0
            nz = 0
            for r in range(n_labels):
                d, n = dt.table[r][r], nt.table[r][r]
                aun = n / (d + 1e-9)
                nz += 1 if 0 < d else 0
                s += aun
                if 0 < d:
                    rec += 1.0 / aun if 0.0 < aun else 1.0e9
                f.write("%s\t%s\t%s\t%s\t%.2f\n" % (prefix, n1, n2, rowname(r), aun))
            f.write("%s\t%s\t%s\tAVERAGE\t%s\n" % (prefix, n1, n2, _na(0 < nz, s / nz if 0 < nz else 0.0)))
            f.write("%s\t%s\t%s\tHARMONIC_MEAN\t%s\n" % (prefix, n1, n2, _na(0 < nz, float(nz) / rec if 0 < nz else 0.0)))
