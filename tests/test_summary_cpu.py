"""Prediction summary tables: the product's hfs_write_all_tables (flagger_amd/csrc/hf_summary.cpp,
through the C ABI) against the literal restatement of summary_table.c in oracle/summary_tables.py — the three files
must be identical byte for byte.  Host-only code: no GPU needed."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from flagger_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import summary_tables as oracle_tables  # noqa: E402


def _make_input(seed, n_contigs=3, n_regions=3, n_annotations=4, n_labels=4, window_len=100, with_truth=True,
                unknown_frac=0.05, run=12):
    """Windows of several contigs cut into chunks; sticky truth / prediction labels; region runs; overlapping annotations."""
    rng = np.random.default_rng(seed)
    chunk_off, cs, ce, ctg = [0], [], [], []
    annot, truth, pred = [], [], []
    for c in range(n_contigs):
        length = int(rng.integers(2_000, 9_000))
        pos = 0
        while pos < length:                               # chunks of 2-3 kb, the last window of a chunk may be short
            end = min(length - 1, pos + int(rng.integers(2_000, 3_000)) - 1)
            n = -(-(end - pos + 1) // window_len)
            cs.append(pos); ce.append(end); ctg.append(("contig_%d" % c).encode())
            chunk_off.append(chunk_off[-1] + n)
            pos = end + 1
    nwin = chunk_off[-1]

    def sticky(n_values, stay):
        out = np.empty(nwin, dtype=np.int64)
        cur = int(rng.integers(0, n_values))
        for i in range(nwin):
            if rng.random() > stay:
                cur = int(rng.integers(0, n_values))
            out[i] = cur
        return out
    region = sticky(n_regions, 1 - 1.0 / (3 * run))
    bits = np.zeros(nwin, dtype=np.uint64)
    for a in range(1, n_annotations):                     # annotation index a <-> bit a-1; index 0 = no annotation bit set
        on = sticky(2, 1 - 1.0 / run).astype(np.uint64)
        bits |= on << np.uint64(a - 1)
    annot = bits | (region.astype(np.uint64) << np.uint64(58))
    truth = sticky(n_labels, 1 - 1.0 / run).astype(np.int8)
    pred = np.where(rng.random(nwin) < 0.7, truth, sticky(n_labels, 1 - 1.0 / run)).astype(np.int8)
    truth[rng.random(nwin) < unknown_frac] = -1
    pred[rng.random(nwin) < unknown_frac] = -1
    names = [b"no_annotation"] + [("annot_%d" % a).encode() for a in range(1, n_annotations)]
    return dict(chunk_off=np.asarray(chunk_off, dtype=np.int64), chunk_s=np.asarray(cs, dtype=np.int32),
                chunk_e=np.asarray(ce, dtype=np.int32), chunk_ctg=ctg, window_len=window_len, annot=annot,
                truth=truth if with_truth else None, prediction=pred, truth_available=int(with_truth),
                prediction_available=1, n_labels=n_labels, n_regions=n_regions, annotation_names=names)


def _run_product(inp, out_path, bins=None, labels=None, thr=0.4, threads=3):
    L = N.lib()
    s = N.hfs_input()
    keep = []
    s.n_windows = int(inp["chunk_off"][-1]); s.n_chunks = len(inp["chunk_s"])
    s.chunk_off = inp["chunk_off"].ctypes.data_as(C.POINTER(C.c_int64))
    s.chunk_s = inp["chunk_s"].ctypes.data_as(C.POINTER(C.c_int32))
    s.chunk_e = inp["chunk_e"].ctypes.data_as(C.POINTER(C.c_int32))
    ctg = (C.c_char_p * len(inp["chunk_ctg"]))(*inp["chunk_ctg"]); keep.append(ctg)
    s.chunk_ctg = ctg
    s.window_len = inp["window_len"]
    s.annot = inp["annot"].ctypes.data_as(C.POINTER(C.c_uint64))
    s.truth = inp["truth"].ctypes.data_as(C.POINTER(C.c_int8)) if inp["truth"] is not None else None
    s.prediction = inp["prediction"].ctypes.data_as(C.POINTER(C.c_int8))
    s.truth_available, s.prediction_available = inp["truth_available"], inp["prediction_available"]
    s.n_labels, s.n_regions, s.n_annotations = inp["n_labels"], inp["n_regions"], len(inp["annotation_names"])
    ann = (C.c_char_p * len(inp["annotation_names"]))(*inp["annotation_names"]); keep.append(ann)
    s.annotation_names = ann
    lab = (C.c_char_p * len(labels))(*[l.encode() for l in labels]) if labels else None
    rc = L.hfs_write_all_tables(C.byref(s), out_path.encode(), bins.encode() if bins else None, lab, len(labels) if labels else 0,
                                thr, threads)
    return rc, L.hfs_last_error().decode()


def _oracle_input(inp):
    d = dict(inp)
    d["chunk_ctg"] = [c.decode() for c in inp["chunk_ctg"]]
    d["annotation_names"] = [a.decode() for a in inp["annotation_names"]]
    d["chunk_off"] = [int(v) for v in inp["chunk_off"]]
    d["chunk_s"] = [int(v) for v in inp["chunk_s"]]
    d["chunk_e"] = [int(v) for v in inp["chunk_e"]]
    return d


SUFFIXES = [".tsv", ".benchmarking.tsv", ".benchmarking.auN_ratio.tsv"]


def _compare(tmp_path, inp, bins=None, labels=None, thr=0.4):
    (tmp_path / "p").mkdir(exist_ok=True); (tmp_path / "o").mkdir(exist_ok=True)
    pp, op = str(tmp_path / "p" / "prediction_summary_final.tsv"), str(tmp_path / "o" / "prediction_summary_final.tsv")
    rc, err = _run_product(inp, pp, bins, labels, thr)
    assert rc == 0, err
    oracle_tables.write_all_tables(_oracle_input(inp), op, bins, labels, thr)
    both = inp["truth_available"] and inp["prediction_available"]
    for suf in SUFFIXES:
        a, b = pp[:-4] + suf, op[:-4] + suf
        assert os.path.exists(a) == os.path.exists(b) == (suf == ".tsv" or bool(both)), suf
        if os.path.exists(a):
            ta, tb = open(a).read(), open(b).read()
            assert ta == tb, (suf, [(x, y) for x, y in zip(ta.splitlines(), tb.splitlines()) if x != y][:3])
    return open(pp).read()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tables_with_truth_match_the_restatement(seed, tmp_path):
    txt = _compare(tmp_path, _make_input(seed), labels=["Err", "Dup", "Hap", "Col", "Unk"])
    assert txt.startswith("#Statistic\tMetric_Type\tEntry_Type\tCategory_Type\tCategory_Name\tSize_Bin_Name\tRef_Label\tErr\tDup\tHap\tCol\tUnk\n")
    assert "TRUTH_VS_PREDICTION\ttruth_based_auN\tcount\tannotation\tannot_2\tALL_SIZES\tCol\t" in txt


def test_without_label_names_rows_are_numbered_and_the_header_has_no_label_columns(tmp_path):
    txt = _compare(tmp_path, _make_input(7, n_regions=1, n_annotations=2))
    assert txt.splitlines()[0].endswith("Ref_Label")            # summary_table.c:1419-1434 with labelNames == NULL
    assert "\tALL_SIZES\t3\t" in txt


def test_prediction_only_has_the_prediction_tables_and_no_benchmarking_files(tmp_path):
    txt = _compare(tmp_path, _make_input(11, with_truth=False), labels=["Err", "Dup", "Hap", "Col", "Unk"])
    kinds = {l.split("\t")[0] for l in txt.splitlines()[1:]}
    assert kinds == {"PREDICTION"}
    assert {l.split("\t")[1] for l in txt.splitlines()[1:]} == {"overlap_based", "base_level"}


def test_size_bins_from_file_overlapping_and_with_gaps(tmp_path):
    bins = tmp_path / "bins.tsv"
    bins.write_text("#start\tend\tname\n0\t300\tshort\n200\t1e3\tmid and a comment after a space\n2000\t1e9\tlong\n")
    txt = _compare(tmp_path, _make_input(5, run=6), bins=str(bins), labels=["Err", "Dup", "Hap", "Col", "Unk"], thr=0.25)
    assert "\tmid\t" in txt and "\tmid and" not in txt          # first space-delimited token of the line (common.c:620-642)


def test_label_name_count_must_match(tmp_path):
    rc, err = _run_product(_make_input(3), str(tmp_path / "x.tsv"), labels=["A", "B", "Unk"])
    assert rc != 0 and "does not match the number of labels" in err


def test_single_window_and_empty_inputs(tmp_path):
    inp = _make_input(4, n_contigs=1)
    one = dict(inp)
    one["chunk_off"] = np.asarray([0, 1], dtype=np.int64); one["chunk_s"] = inp["chunk_s"][:1]; one["chunk_e"] = inp["chunk_s"][:1] + 49
    one["chunk_ctg"] = inp["chunk_ctg"][:1]
    for k in ("annot", "truth", "prediction"):
        one[k] = inp[k][:1].copy()
    _compare(tmp_path, one, labels=["Err", "Dup", "Hap", "Col", "Unk"])
    empty = dict(one)
    empty["chunk_off"] = np.asarray([0], dtype=np.int64); empty["chunk_s"] = inp["chunk_s"][:0]; empty["chunk_e"] = inp["chunk_e"][:0]
    empty["chunk_ctg"] = []
    for k in ("annot", "truth", "prediction"):
        empty[k] = inp[k][:0].copy()
    _compare(tmp_path, empty, labels=["Err", "Dup", "Hap", "Col", "Unk"])
