"""Known answers that the REFERENCE's own test-suite holds for the code either side of the hot path,
reproduced with the reference's data files (copied verbatim into tests/golden/) by BOTH the oracle and the product:

* programs/tests/test_summary_table.c:12-270   SummaryTable / SummaryTableList counts, percentages, "%.2f" row strings
* programs/tests/test_summary_table.c:272-470  the 5x5 PREDICTION_VS_TRUTH tables (base-level and overlap-based) of
                                               tests/test_files/summary_table/test_1.cov for whole_genome, annotation_1,
                                               annotation_2 and the two size bins of test_1_bin_array.txt
* programs/tests/test_common.c:22-130          "%.5e" joins, IntBinArray parsing and (overlapping) bin lookup
* programs/tests/test_track_reader.c:9-60      the rows of tests/test_files/track_reader/test_1.cov{,.gz}
* programs/tests/test_track_reader.c:63-205    header fields written by attributes and read back
* programs/tests/test_chunk_iterator.c:9-45    windows (coordinates, values, region, annotations, labels) at chunkLen 40,
                                               windowLen 20 of tests/test_files/chunk_iterator/test_1_with_labels.cov

The expected numbers below are the literals of those test files.  Nothing here needs a GPU."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from flagger_amd import _native as N
from flagger_amd import hmm, synth
from flagger_amd import io as fio
from test_oracle_cpu import GOLD, ROOT, _oracle_load_cov
from test_summary_cpu import _run_product

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import summary_tables as oracle_tables  # noqa: E402


# ---------------------------------------------------------------- test_summary_table.c:12-270
def test_summary_table_increment_and_row_strings():
    t = oracle_tables.SummaryTable(2, 2)
    for r, c, v in ((0, 0, 10), (0, 0, 10), (0, 0, 10), (0, 1, 10), (0, 1, 10)):
        t.increment(r, c, v)
    assert t.table == [[30, 20], [0, 0]] and t.pct == [[60, 40], [0, 0]] and t.row_total == [50, 0]
    fmt = lambda vals: ",".join("%.2f" % v for v in vals)   # SummaryTable_getRowString, summary_table.c:118-131
    assert fmt(t.table[0]) == "30.00,20.00" and fmt(t.table[1]) == "0.00,0.00"
    assert fmt(t.pct[0]) == "60.00,40.00" and fmt(t.pct[1]) == "0.00,0.00"
    assert "TEST," + fmt(t.row_total) == "TEST,50.00,0.00" and "TEST," + fmt(t.row_total_pct) == "TEST,100.00,0.00"
    # SummaryTableList: (cat1, cat2) = (1, 1) of test_SummaryTableList_getRowString
    u = oracle_tables.SummaryTable(2, 2)
    for r, c, v in ((0, 0, 40), (0, 0, 20), (0, 1, 10), (0, 1, 10), (1, 1, 20)):
        u.increment(r, c, v)
    assert fmt(u.table[0]) == "60.00,20.00" and fmt(u.table[1]) == "0.00,20.00"
    assert fmt(u.pct[0]) == "75.00,25.00" and fmt(u.pct[1]) == "0.00,100.00"
    assert fmt(u.row_total) == "80.00,20.00" and fmt(u.row_total_pct) == "80.00,20.00"


def _expected_tables(overlap):
    """test_summary_table.c:275-360: [annotation index][size bin] -> 5x5 (rows = prediction label, columns = truth label)."""
    z = lambda: np.zeros((5, 5))
    wg1, wg2, a11, a12, a21, a22 = z(), z(), z(), z(), z(), z()
    if overlap:
        wg1[1, 3] = 1; wg1[4, 1] = 1; wg1[4, 4] = 3
        wg2[0, 0] = 1; wg2[2, 2] = 1; wg2[2, 4] = 1; wg2[3, 1] = 1; wg2[3, 3] = 1; wg2[4, 0] = 1; wg2[4, 4] = 3
        a11[0, 0] = 1; a11[4, 0] = 1; a11[4, 4] = 1
        a12[2, 4] = 1
        a21[1, 3] = 1; a21[4, 4] = 2
        a22[2, 2] = 1; a22[2, 4] = 1; a22[3, 1] = 1; a22[3, 3] = 1
    else:
        wg1[1, 3] = 2; wg1[4, 1] = 1; wg1[4, 4] = 5
        wg2[0, 0] = 2; wg2[0, 4] = 1; wg2[2, 1] = 1; wg2[2, 2] = 6; wg2[2, 4] = 6; wg2[3, 0] = 1; wg2[3, 1] = 4; wg2[3, 3] = 3
        wg2[4, 0] = 4; wg2[4, 2] = 1; wg2[4, 4] = 13
        a11[0, 0] = 2; a11[4, 0] = 2; a11[4, 4] = 1
        a12[2, 1] = 1; a12[2, 4] = 2
        a21[1, 3] = 2; a21[4, 4] = 3
        a22[2, 2] = 4; a22[2, 4] = 4; a22[3, 1] = 4; a22[3, 3] = 3
    return {1: [wg1, wg2], 2: [a11, a12], 3: [a21, a22]}


def _summary_input(store):
    return dict(chunk_off=np.asarray(store.chunk_off, np.int64), chunk_s=np.asarray(store.chunk_s, np.int32),
                chunk_e=np.asarray(store.chunk_e, np.int32), chunk_ctg=[c.encode() for c in store.chunk_ctg], window_len=store.window_len,
                annot=np.asarray(store.annot, np.uint64), truth=np.asarray(store.truth, np.int8),
                prediction=np.asarray(store.prediction, np.int8), truth_available=1, prediction_available=1, n_labels=4,
                n_regions=store.n_regions, annotation_names=[a.encode() for a in store.annotation_names])


def _tables_from_tsv(path, comparison, metric, category="annotation"):
    """{(category name, bin name): {row label: [values]}} of the count rows of a prediction_summary TSV."""
    out = {}
    for line in open(path):
        if line.startswith("#"):
            continue
        t = line.rstrip("\n").split("\t")
        if t[0] == comparison and t[1] == metric and t[2] == "count" and t[3] == category:
            out.setdefault((t[4], t[5]), {})[t[6]] = [float(v) for v in t[7:]]
    return out


@pytest.mark.parametrize("overlap", [False, True], ids=["base_level", "overlap_based"])
@pytest.mark.parametrize("loader", ["product", "oracle"])
def test_summary_tables_of_the_reference_fixture(overlap, loader, tmp_path):
    cov = os.path.join(GOLD, "summary_table_test_1.cov")
    bins = os.path.join(GOLD, "summary_table_test_1_bin_array.txt")
    # windowLen 1, chunkCanonicalLen 10 as in test_summary_table.c:388-396 (the block iterator of tests 6 / 8 gives the same tables)
    store = fio.Table(cov, 10, 1).store() if loader == "product" else _oracle_load_cov(cov, 10, 1, tmp_path)
    assert list(store.annotation_names) == ["no_annotation", "whole_genome", "annotation_1", "annotation_2"]
    assert store.n_windows == 50 and (store.truth >= -1).all() and (store.prediction >= -1).all()
    expected = _expected_tables(overlap)
    metric = oracle_tables.OVERLAP_BASED if overlap else oracle_tables.BASE_LEVEL
    inp = _summary_input(store)
    # 1. the oracle's updater, table by table (overlapRatioThreshold 0.4, test_summary_table.c:417)
    oinp = dict(inp, chunk_ctg=list(store.chunk_ctg), annotation_names=list(store.annotation_names),
                chunk_off=[int(v) for v in store.chunk_off], chunk_s=[int(v) for v in store.chunk_s], chunk_e=[int(v) for v in store.chunk_e])
    b = oracle_tables.read_bins(bins)
    assert b == ([0, 3], [3, 100], ["0-2", "2<"])
    tl = oracle_tables.TableList(list(store.annotation_names), b[2], 5)
    for c1 in range(4):
        oracle_tables.update(oinp, tl, None, b, oracle_tables.CAT_ANNOTATION, c1, metric, oracle_tables.PRED_VS_TRUTH, 0.4)
    for c1, per_bin in expected.items():
        for c2 in range(2):
            assert np.array_equal(np.asarray(tl.tabs[c1][c2].table), per_bin[c2]), (c1, c2)
    # 2. the product's writer (hf_summary.cpp through the C ABI) and the oracle's writer: the same numbers in the file
    for who in ("product", "oracle"):
        out = str(tmp_path / (who + ".tsv"))
        if who == "product":
            _run_product(inp, out, bins=bins, thr=0.4)
        else:
            oracle_tables.write_all_tables(oinp, out, bins, None, 0.4)
        got = _tables_from_tsv(out, "PREDICTION_VS_TRUTH", "overlap_based" if overlap else "base_level")
        names = {1: "whole_genome", 2: "annotation_1", 3: "annotation_2"}
        for c1, per_bin in expected.items():
            for c2, bn in enumerate(("0-2", "2<")):
                tab = got[(names[c1], bn)]
                for r in range(5):
                    assert tab["%d" % r] == list(per_bin[c2][r]), (who, c1, bn, r)
    assert open(str(tmp_path / "product.tsv")).read() == open(str(tmp_path / "oracle.tsv")).read()


def test_summary_tables_with_an_all_sizes_bin_and_prediction_only_labels(tmp_path):
    """test_1_bin_array_with_all.txt adds a bin that contains the other two (IntBinArray_getBinIndices returns every bin a
    length falls into, test_common.c:109-127): its table is the sum of theirs.  test_1.only_prediction_labels.cov has no
    truth column: only the PREDICTION tables are written, by product and oracle alike."""
    bins = os.path.join(GOLD, "summary_table_test_1_bin_array_with_all.txt")
    store = fio.Table(os.path.join(GOLD, "summary_table_test_1.cov"), 10, 1).store()
    out = str(tmp_path / "all.tsv")
    _run_product(_summary_input(store), out, bins=bins, thr=0.4)
    for overlap in (False, True):
        got = _tables_from_tsv(out, "PREDICTION_VS_TRUTH", "overlap_based" if overlap else "base_level")
        exp = _expected_tables(overlap)
        for c1, name in ((1, "whole_genome"), (2, "annotation_1"), (3, "annotation_2")):
            for r in range(5):
                assert got[(name, "ALL_SIZES")]["%d" % r] == list(exp[c1][0][r] + exp[c1][1][r])
                assert got[(name, "0-2")]["%d" % r] == list(exp[c1][0][r])
    po = fio.Table(os.path.join(GOLD, "summary_table_test_1.only_prediction_labels.cov"), 10, 1).store()
    inp = dict(_summary_input(po), truth=None, truth_available=0)
    _run_product(inp, str(tmp_path / "po.tsv"), bins=bins, thr=0.4)
    oinp = dict(inp, chunk_ctg=list(po.chunk_ctg), annotation_names=list(po.annotation_names), chunk_off=[int(v) for v in po.chunk_off],
                chunk_s=[int(v) for v in po.chunk_s], chunk_e=[int(v) for v in po.chunk_e])
    oracle_tables.write_all_tables(oinp, str(tmp_path / "po_ref.tsv"), bins, None, 0.4)
    txt = open(str(tmp_path / "po.tsv")).read()
    assert txt == open(str(tmp_path / "po_ref.tsv")).read()
    assert "PREDICTION\t" in txt and "TRUTH" not in txt.replace("PREDICTION", "")
    assert not os.path.exists(str(tmp_path / "po.benchmarking.tsv"))


# ---------------------------------------------------------------- test_common.c:22-130
def test_double_join_format_and_bin_array_parsing(tmp_path):
    # String_joinDoubleArray: "%.5e" joined by ',' (test_common.c:22-31) — the format of emission_*.tsv (hmm_utils.c:1539-1575)
    store = synth.config(1, scale=0.05)
    model = hmm.createModel(hmm.MODEL_GAUSSIAN, 3, store, np.zeros((4, 4)))
    v = model.param_vector().reshape(1, -1)
    K = N.HF_MAXCOMP
    v[0, 27 + 3 * K:27 + 3 * K + 3] = [1e-3, 0.1, 2.33]          # means of the collapsed state's three components
    model.set_param_vector(v.ravel())
    p = str(tmp_path / "emission.tsv")
    model.writeEmissionTsv(p)
    rows = [l.rstrip("\n").split("\t") for l in open(p) if l.startswith("Col")]
    mean_row = [r for r in rows if r[3] == "Mean"][0]
    assert mean_row[4] == "1.00000e-03,1.00000e-01,2.33000e+00"
    # IntBinArray_constructFromFile (test_common.c:84-105): "200\t1e9\t200<" parses through atof
    b = oracle_tables.read_bins(os.path.join(GOLD, "common_bin_array.txt"))
    assert b == ([0, 100, 200], [100, 200, 1000000000], ["0-100", "100-200", "200<"])
    index = lambda bb, x: [i for i in range(len(bb[0])) if bb[0][i] <= x < bb[1][i]]
    assert [index(b, x) for x in (0, 50, 100, 200, 1000)] == [[0], [0], [1], [2], [2]]
    # overlapping bins (test_common.c:109-127): 50 -> bins 0 and 1, 80 -> bin 1 only
    b2 = oracle_tables.read_bins(os.path.join(GOLD, "common_bin_array_2.txt"))
    assert b2[2] == ["0-60", "0-100", "100-200", "200<"] and index(b2, 50) == [0, 1] and index(b2, 80) == [1]
    # the product parses the same files: one 50-base and one 80-base block of label 0 land in the bins the reference says
    ann = np.full(130, 1, dtype=np.uint64)
    inp = dict(chunk_off=np.asarray([0, 50, 130], np.int64), chunk_s=np.asarray([0, 0], np.int32), chunk_e=np.asarray([49, 79], np.int32),
               chunk_ctg=[b"a", b"b"], window_len=1, annot=ann, truth=None, prediction=np.zeros(130, np.int8), truth_available=0,
               prediction_available=1, n_labels=2, n_regions=1, annotation_names=[b"no_annotation", b"all"])
    out = str(tmp_path / "bins.tsv")
    _run_product(inp, out, bins=os.path.join(GOLD, "common_bin_array_2.txt"))
    rows = {}
    for line in open(out):
        t = line.rstrip("\n").split("\t")
        if t[:5] == ["PREDICTION", "overlap_based", "count", "annotation", "all"]:
            rows[t[5]] = [float(x) for x in t[7:]]
    assert rows == {"0-60": [1.0, 0.0, 0.0], "0-100": [2.0, 0.0, 0.0], "100-200": [0.0, 0.0, 0.0], "200<": [0.0, 0.0, 0.0]}


# ---------------------------------------------------------------- test_track_reader.c:9-60, 100-205
TRACKS = {"ctg1": [(1, 10, 4, 4, 4, 0), (11, 15, 6, 0, 0, 0), (16, 20, 6, 0, 0, 1), (21, 60, 10, 10, 10, 1), (61, 64, 14, 14, 10, 1),
                   (65, 70, 14, 14, 10, 0), (71, 110, 16, 16, 16, 1)],
          "ctg2": [(1, 2, 4, 4, 4, 1), (3, 10, 8, 8, 0, 0)]}


@pytest.mark.parametrize("fname", ["track_reader_test_1.cov", "track_reader_test_1.cov.gz"])
@pytest.mark.parametrize("loader", ["product", "oracle"])
def test_track_reader_rows(fname, loader, tmp_path):
    """One window per base (windowLen 1) turns the loader's output back into the rows TrackReader_next yields.
    The reference's test_1.cov.gz still carries a pre-1.0 header ('#region:0:5' for '#region:coverage:0:5'): TrackReader reads
    its rows, but CoverageHeader_construct — what hmm_flagger runs first — exits with 'Number of parsed region coverages ...'
    (track_reader.c:356-372).  Product and oracle refuse the file as is with that message, and give the reference's rows once
    the two header lines are spelled the current way."""
    import gzip
    import shutil
    p = tmp_path / fname.replace("track_reader_", "")
    shutil.copy(os.path.join(GOLD, fname), p)
    load = (lambda q: fio.Table(str(q), 10000, 1).store()) if loader == "product" else (lambda q: _oracle_load_cov(str(q), 10000, 1, tmp_path))
    if fname.endswith(".gz"):
        with pytest.raises(Exception) as ei:
            load(p)
        if loader == "product":
            assert "Number of parsed region coverages" in str(ei.value)
        text = gzip.open(p, "rt").read().replace("#region:0:", "#region:coverage:0:").replace("#region:1:", "#region:coverage:1:")
        with gzip.open(p, "wt") as f:
            f.write(text)
    store = load(p)
    assert store.region_coverages == [5, 10] and list(store.annotation_names) == ["NO_ANNOTATION", "annotation_1", "annotation_2"]
    for c in range(store.n_chunks):
        a = int(store.chunk_off[c])
        assert int(store.chunk_s[c]) == 0
        for s, e, cov, mapq, clip, region in TRACKS[store.chunk_ctg[c]]:
            sl = slice(a + s - 1, a + e)
            assert (store.cov[sl] == cov).all() and (store.mapq[sl] == mapq).all() and (store.clip[sl] == clip).all()
            assert ((store.annot[sl] >> np.uint64(58)) == region).all()
        assert int(store.chunk_off[c + 1]) - a == TRACKS[store.chunk_ctg[c]][-1][1]


def test_coverage_header_fields_by_attributes(tmp_path):
    """CoverageHeader_constructByAttributes -> write -> read (test_track_reader.c:131-205): two annotations, three regions
    with coverages 5, 10, 25, no labels, no truth / prediction.  The header text is what track_reader.c:110-202 writes."""
    p = tmp_path / "h.cov"
    p.write_text("#annotation:len:2\n#annotation:name:0:no_annotation\n#annotation:name:1:annotation_1\n#region:len:3\n"
                 "#region:coverage:0:5\n#region:coverage:1:10\n#region:coverage:2:25\n#label:len:0\n#truth:false\n#prediction:false\n"
                 "#avg_alignment_len:0\n#start-only:false\n>c 4\n1\t4\t3\t3\t0\t1\t2\n")
    t = fio.Table(str(p), 100, 2)
    st = t.store()
    assert list(st.annotation_names) == ["no_annotation", "annotation_1"] and st.region_coverages == [5, 10, 25]
    assert st.n_windows == 2 and (st.truth == -1).all() and (st.prediction == -1).all() and not st.start_only
    assert ((st.annot >> np.uint64(58)) == 2).all()


# ---------------------------------------------------------------- test_chunk_iterator.c:9-45
def test_chunk_iterator_windows(tmp_path):
    cov = os.path.join(GOLD, "chunks_creator_test_1_with_labels.cov")     # = tests/test_files/chunk_iterator/test_1_with_labels.cov
    ctg1 = [(0, 19, 5, 2, 2, 0), (20, 39, 10, 10, 10, 1), (40, 59, 10, 10, 10, 1), (60, 79, 15, 15, 13, 1), (80, 99, 16, 16, 16, 1),
            (100, 109, 16, 16, 16, 1)]
    ctg2 = [(0, 9, 7, 7, 1, 0)]
    ann1, ann2 = [[1], [1], [1], [1, 2], [2], [2]], [[1, 2]]
    lab1, lab2 = [(1, 1), (2, 3), (2, 3), (3, 3), (3, 3), (3, 3)], [(2, 2)]
    for store in (fio.Table(cov, 40, 20).store(), _oracle_load_cov(cov, 40, 20, tmp_path)):
        rows = {"ctg1": [], "ctg2": []}
        for c in range(store.n_chunks):
            a, b = int(store.chunk_off[c]), int(store.chunk_off[c + 1])
            for i in range(a, b):                                         # ChunkIterator_getNextPtBlock, chunk.c:915-950
                s = int(store.chunk_s[c]) + 20 * (i - a)
                e = min(int(store.chunk_e[c]), s + 19)
                bits = int(store.annot[i]) & ~(0x3F << 58)
                rows[store.chunk_ctg[c]].append(((s, e, int(store.cov[i]), int(store.mapq[i]), int(store.clip[i]), int(store.annot[i]) >> 58),
                                                 [k + 1 for k in range(58) if bits >> k & 1],
                                                 (int(store.truth[i]), int(store.prediction[i]))))
        assert [r[0] for r in rows["ctg1"]] == ctg1 and [r[0] for r in rows["ctg2"]] == ctg2
        assert [r[1] for r in rows["ctg1"]] == ann1 and [r[1] for r in rows["ctg2"]] == ann2
        assert [r[2] for r in rows["ctg1"]] == lab1 and [r[2] for r in rows["ctg2"]] == lab2
