"""Host loader / writers (include/hmm_flagger_io.h) on CPU: the reference's own known answers for window
averaging, the oracle's loader and writers as cross-checks, `.bin` round trips."""
import os
import shutil

import numpy as np
import pytest

from flagger_amd import io as fio
from flagger_amd import synth
from oracle_py import Oracle
from test_oracle_cpu import GOLD, _oracle_load_cov, check_chunks_creator_expectations


def _same_store(a, b):
    for f in ("cov", "mapq", "clip", "annot", "truth", "prediction", "chunk_off", "chunk_s", "chunk_e", "chunk_ctg_len"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert a.chunk_ctg == b.chunk_ctg and a.region_coverages == b.region_coverages
    assert (a.window_len, a.chunk_len, a.avg_alignment_len, a.start_only) == (b.window_len, b.chunk_len, b.avg_alignment_len, b.start_only)
    assert list(a.annotation_names) == list(b.annotation_names)


@pytest.mark.parametrize("fname,labels", [("chunks_creator_test_1.cov", False), ("chunks_creator_test_1.cov.gz", False),
                                          ("chunks_creator_test_1_with_labels.cov", True)])
def test_loader_reference_known_answers(fname, labels, tmp_path):
    """programs/tests/test_chunks_creator.c:12-29,126-133 on its own data files."""
    p = tmp_path / fname.replace("chunks_creator_", "")
    shutil.copy(os.path.join(GOLD, fname), p)
    t = fio.Table(str(p), 40, 20)
    store = t.store()
    assert store.region_coverages == [5, 10] and store.n_chunks == 3
    check_chunks_creator_expectations(store, labels)
    _same_store(store, _oracle_load_cov(str(p), 40, 20, tmp_path))
    # chunkLen 20 reproduces the chunk list of the reference's committed index file (test_1.cov.gz.index)
    s20 = fio.Table(str(p), 20, 20).store()
    assert [(c, int(s), int(e)) for c, s, e in zip(s20.chunk_ctg, s20.chunk_s, s20.chunk_e)] == [
        ("ctg1", 0, 19), ("ctg1", 20, 39), ("ctg1", 40, 59), ("ctg1", 60, 79), ("ctg1", 80, 109), ("ctg2", 0, 9)]


def test_loader_matches_oracle_on_simulated_cov(tmp_path):
    p = os.path.join(GOLD, "sim_gaussian_30k.cov.gz")
    for chunk_len, window_len in [(1000, 1), (7000, 13), (50_000, 400)]:
        mine = fio.Table(p, chunk_len, window_len).store()
        _same_store(mine, _oracle_load_cov(p, chunk_len, window_len, tmp_path))
    twin = synth.WindowStore.read_bin(os.path.join(GOLD, "sim_gaussian_30k.bin"))
    _same_store(fio.Table(p, 1000, 1).store(), twin)


def test_loader_run_lengths_fractions_and_start_only(tmp_path):
    """Run-length rows with non-integer values must reproduce the reference's per-base accumulation exactly
    (the oracle loops per base); start-only mode rescales partial windows (chunk.c:398-403)."""
    rng = np.random.default_rng(4)
    for start_only in (False, True):
        lines = ["#annotation:len:2", "#annotation:name:0:no_annotation", "#annotation:name:1:whole_genome",
                 "#region:len:2", "#region:coverage:0:20", "#region:coverage:1:31", "#label:len:4", "#truth:true",
                 "#avg_alignment_len:12000", f"#start-only:{'true' if start_only else 'false'}"]
        for ci, L in enumerate([5003, 777, 20011]):
            lines.append(f">c{ci} {L}")
            pos = 1
            while pos <= L:
                run = int(min(L - pos + 1, rng.integers(1, 400)))
                cov = rng.choice([rng.integers(0, 300), round(float(rng.uniform(0, 60)), 2), 0.1, 1e-3])
                lines.append(f"{pos}\t{pos + run - 1}\t{cov}\t{round(float(rng.uniform(0, 30)), 1)}\t{int(rng.integers(0, 9))}\t"
                             f"{rng.choice(['0', '1', '1,2'])}\t{int(rng.integers(0, 2))}\t{int(rng.integers(-1, 4))}")
                pos += run
        p = tmp_path / f"so{int(start_only)}.cov"
        p.write_text("\n".join(lines) + "\n")
        for chunk_len, window_len in [(2000, 50), (100_000, 7), (3000, 3000)]:
            _same_store(fio.Table(str(p), chunk_len, window_len).store(), _oracle_load_cov(str(p), chunk_len, window_len, tmp_path))


def test_bin_round_trip_and_header(tmp_path):
    store = synth.config(4, scale=0.003)
    p1, p2 = tmp_path / "a.bin", tmp_path / "b.bin"
    store.write_bin(str(p1))
    t = fio.Table(str(p1))
    _same_store(t.store(), store)
    t.write_bin(str(p2))
    assert p1.read_bytes() == p2.read_bytes()


def test_loader_errors(tmp_path):
    with pytest.raises(OSError):
        fio.Table(str(tmp_path / "missing.cov"))
    with pytest.raises(OSError):
        fio.Table(str(tmp_path / "x.txt"))
    bad = tmp_path / "bad.cov"
    bad.write_text("#annotation:len:1\n>c 10\n1\t10\t3\t3\t0\t0\t0\n")     # no #region:len
    with pytest.raises(OSError):
        fio.Table(str(bad))
    gap = tmp_path / "gap.cov"
    gap.write_text("#annotation:len:1\n#region:len:1\n#region:coverage:0:20\n>c 10\n1\t4\t3\t3\t0\t0\t0\n6\t10\t3\t3\t0\t0\t0\n")
    with pytest.raises(OSError):
        fio.Table(str(gap))


@pytest.mark.parametrize("min_len", [(0, 0, 0, 0), (2000, 0, 0, 0), (5000, 9000, 0, 3000)])
def test_final_bed_and_posterior_bed_match_oracle_writers(min_len, tmp_path):
    store = synth.synthesize([400_000, 90_500, 12_000], 1000, 150_000, [20], seed=6)
    rng = np.random.default_rng(1)
    # label runs of random lengths, including -1 (unknown)
    lab = np.repeat(rng.integers(-1, 4, size=200), rng.integers(1, 9, size=200))[:store.n_windows].astype(np.int8)
    assert lab.size == store.n_windows
    binp = tmp_path / "s.bin"
    store.write_bin(str(binp))
    t = fio.Table(str(binp))
    mine = tmp_path / "mine.bed"
    t.write_final_bed(lab, str(mine), "trk", min_len)
    orc = Oracle(store, 0, 3, None)
    off = 0
    for c in range(orc.cc.contents.n_chunks):
        ch = orc.cc.contents.chunks[c]
        for i in range(ch.n):
            ch.prediction[i] = int(lab[off + i])
        off += ch.n
    theirs = tmp_path / "oracle.bed"
    orc.write_final_bed(str(theirs), "trk", min_len)
    assert mine.read_text() == theirs.read_text()
    orc.close()


@pytest.mark.parametrize("ext", ["cov", "cov.gz"])
def test_synthetic_store_written_as_cov_loads_back_identically(ext, tmp_path):
    """WindowStore.write_cov (one run per window, the simulator's header layout) -> product loader and oracle loader:
    the same windows, chunks, regions, annotations and truth labels as the store; --contigsList subsetting."""
    st = synth.synthesize([250_000, 61_000, 1_000], 1000, 100_000, [20, 33], seed=9, region_run_bases=(4_000, 30_000))
    path = str(tmp_path / f"s.{ext}")
    st.write_cov(path)
    tab = fio.Table(path, 100_000, 1000)
    got = tab.store()
    for name in ("cov", "mapq", "clip", "annot", "truth", "chunk_off", "chunk_s", "chunk_e", "chunk_ctg_len"):
        assert np.array_equal(getattr(got, name), getattr(st, name)), name
    assert list(got.chunk_ctg) == list(st.chunk_ctg) and got.region_coverages == st.region_coverages
    ref = _oracle_load_cov(path, 100_000, 1000, tmp_path)
    assert np.array_equal(ref.cov, st.cov) and np.array_equal(ref.annot, st.annot) and np.array_equal(ref.chunk_off, st.chunk_off)
    # keep two of the three contigs (hfio_subset_contigs, chunk.c:218-237)
    import ctypes as C
    L = tab._L
    L.hfio_subset_contigs.restype = C.c_int32
    L.hfio_subset_contigs.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_int]
    names = (C.c_char_p * 2)(b"ctg2", b"ctg0")
    kept = L.hfio_subset_contigs(tab._h, names, 2)
    sub = tab.store()
    want = st.subset_chunks([c for c in range(st.n_chunks) if st.chunk_ctg[c] in ("ctg0", "ctg2")])
    assert kept == want.n_chunks and np.array_equal(sub.cov, want.cov) and list(sub.chunk_ctg) == list(want.chunk_ctg)


def test_loader_number_shapes_line_endings_and_block_boundaries(tmp_path):
    """The block reader (one thread inflates line-aligned blocks of ~4 MiB, a few threads turn their lines into records, the
    caller consumes them in file order) and the fast field parser against the oracle's gzgets + atoi/atof: every number shape falls back to the library where the fast path does not apply; lines
    that straddle block boundaries; a last line without newline; CRLF line endings."""
    header = ["#annotation:len:2", "#annotation:name:0:no_annotation", "#annotation:name:1:whole_genome", "#region:len:1",
              "#region:coverage:0:20", "#label:len:4", "#truth:true", "#avg_alignment_len:9000", "#start-only:false"]
    shapes = ["12", "12.", "012.50", "1e1", "1.5E+1", "+7", ".5", "  8", "3.141592653589793238", "0.1", "7.25", "1234567.125",
              "0", "00", "249.99", "250.5", "1e-3", "123456789012345", "1234567890123456", "9.999999999999999", "nan"]
    rng = np.random.default_rng(11)
    lines = list(header) + [">c0 400000"]
    for i in range(400_000):                       # one base per row: ~9 MB of text, two block boundaries
        v = shapes[i % len(shapes)] if i % 7 == 0 and shapes[i % len(shapes)] != "nan" else str(int(rng.integers(0, 60)))
        lines.append(f"{i + 1}\t{i + 1}\t{v}\t{int(rng.integers(0, 30))}.5\t{int(rng.integers(0, 5))}\t1\t0\t{int(rng.integers(-1, 4))}")
    text = "\n".join(lines) + "\n"
    assert len(text) > 2 * (4 << 20)
    p = tmp_path / "shapes.cov"
    p.write_text(text)
    ref = _oracle_load_cov(str(p), 50_000, 100, tmp_path)
    mine = fio.Table(str(p), 50_000, 100).store()
    _same_store(mine, ref)
    (tmp_path / "nonl.cov").write_text(text[:-1])                      # no newline after the last row
    _same_store(fio.Table(str(tmp_path / "nonl.cov"), 50_000, 100).store(), ref)
    (tmp_path / "crlf.cov").write_text(text.replace("\n", "\r\n"))
    _same_store(fio.Table(str(tmp_path / "crlf.cov"), 50_000, 100).store(), ref)
    import gzip
    with gzip.open(tmp_path / "z.cov.gz", "wt") as f:
        f.write(text)
    _same_store(fio.Table(str(tmp_path / "z.cov.gz"), 50_000, 100).store(), ref)
    # a line longer than a block (a 5 MiB comment in the header): the block grows until it holds a line end
    long_line = "#" + "x" * (5 << 20)
    (tmp_path / "long.cov").write_text("\n".join(header[:3] + [long_line] + header[3:] + lines[len(header):]) + "\n")
    _same_store(fio.Table(str(tmp_path / "long.cov"), 50_000, 100).store(), ref)
    (tmp_path / "empty.cov").write_text("")
    with pytest.raises(Exception):
        fio.Table(str(tmp_path / "empty.cov"), 50_000, 100)


def test_truncated_gz_is_an_error(tmp_path):
    """A .cov.gz cut anywhere before its trailer must be refused, not loaded as a shorter genome (zlib reports
    Z_BUF_ERROR / Z_DATA_ERROR at the cut; the cut may fall after a complete row)."""
    src = open(os.path.join(GOLD, "sim_gaussian_30k.cov.gz"), "rb").read()
    ok = fio.Table(os.path.join(GOLD, "sim_gaussian_30k.cov.gz"), 1000, 1).store()
    assert ok.n_windows == 30000
    for cut in (len(src) // 3, len(src) // 2, len(src) - 9, len(src) - 1):
        p = tmp_path / f"cut_{cut}.cov.gz"
        p.write_bytes(src[:cut])
        with pytest.raises(Exception):
            fio.Table(str(p), 1000, 1)
    # garbage in the middle of the deflate stream
    bad = bytearray(src)
    for k in range(len(bad) // 2, len(bad) // 2 + 64):
        bad[k] ^= 0xA5
    p = tmp_path / "corrupt.cov.gz"
    p.write_bytes(bytes(bad))
    with pytest.raises(Exception):
        fio.Table(str(p), 1000, 1)


def test_loader_at_real_row_density_matches_the_oracle_loader(tmp_path):
    """a bam2cov-like track changes value every few hundred bases (synth.write_cov_dense: runs of 50-500 bases, ~15 rows per
    4 kb window, every window and chunk boundary straddled by a run), where synth.WindowStore.write_cov has one run per window.  Reduced size
    here (3.2 Mb, 11 k rows); the full-size run (51 M rows, 1.5 GB of text in one DEFLATE stream) is profiles/r06_loader_dense.txt.
    The product loader against the oracle's per-base loop (chunk.c:393-483, track_reader.c:751-818) for several chunk / window lengths,
    and a single-contig file (`only`) against the same contig of the full file — what the full-size run samples."""
    lengths = [2_511_003, 611_999, 40_004, 3_000]
    full = str(tmp_path / "dense.cov.gz")
    info = synth.write_cov_dense(full, lengths, seed=11, min_run=50, max_run=500)
    assert info["bases"] == sum(lengths) and 9_000 < info["rows"] < 16_000
    for chunk_len, window_len in [(1_000_000, 4000), (300_000, 16000), (2_000_000, 777)]:
        mine = fio.Table(full, chunk_len, window_len).store()
        _same_store(mine, _oracle_load_cov(full, chunk_len, window_len, tmp_path))
        assert mine.n_windows >= sum((L + window_len - 1) // window_len for L in lengths) - 8
    # the plain-text twin (no DEFLATE) and zlib's reader give the same windows
    txt = str(tmp_path / "dense.cov")
    assert synth.write_cov_dense(txt, lengths, seed=11, min_run=50, max_run=500) == info
    _same_store(fio.Table(txt, 1_000_000, 4000).store(), fio.Table(full, 1_000_000, 4000).store())
    # contig 1 alone: the rows it has in the full file
    one = str(tmp_path / "c1.cov.gz")
    synth.write_cov_dense(one, lengths, seed=11, min_run=50, max_run=500, only=1)
    a, b = fio.Table(one, 1_000_000, 4000).store(), fio.Table(full, 1_000_000, 4000).store()
    sel = [c for c in range(b.n_chunks) if b.chunk_ctg[c] == "hap_ctg1"]
    want = b.subset_chunks(sel)
    assert a.n_chunks == len(sel) and np.array_equal(a.cov, want.cov) and np.array_equal(a.mapq, want.mapq) and np.array_equal(a.clip, want.clip)
    assert np.array_equal(a.annot, want.annot) and list(a.chunk_ctg) == list(want.chunk_ctg)


def _load_in_subprocess(path, env, chunk_len=1_000_000, window_len=4000):
    """(digest of the loaded store, the loader's trace line) from a fresh process with `env` (the loader reads its switches once per file)."""
    import subprocess
    import sys
    code = ("import sys, hashlib\nsys.path.insert(0, %r)\nfrom flagger_amd import io as fio\n"
            "try:\n    st = fio.Table(%r, %d, %d).store()\nexcept OSError as e:\n    print('ERROR', e); sys.exit(0)\n"
            "h = hashlib.sha256()\n"
            "for f in ('cov', 'mapq', 'clip', 'annot', 'truth', 'chunk_off', 'chunk_s', 'chunk_e', 'chunk_ctg_len'):\n    h.update(getattr(st, f).tobytes())\n"
            "print('OK', st.n_windows, st.n_chunks, h.hexdigest())\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, chunk_len, window_len)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, HF_IO_TRACE="1", **env))
    assert r.returncode == 0, r.stderr[-2000:]
    trace = [l for l in r.stderr.splitlines() if l.startswith("[hfio]")]
    return r.stdout.strip(), (trace[-1] if trace else "")


def _rounds_and_fallbacks(trace):
    import re
    m = re.search(r"(\d+) rounds of parallel decoders, (\d+) fall-backs", trace)
    return (int(m.group(1)), int(m.group(2))) if m else (0, 0)


SPEC = {"HF_IO_PARALLEL": "4", "HF_IO_PARALLEL_MIN": "0", "HF_IO_PIECE": "40000", "HF_IO_PROBE": "100000"}


def test_parallel_decoders_inside_one_deflate_stream(tmp_path):
    """Round 6: a `.cov.gz` as the reference writes it (gzopen "w6h": Z_HUFFMAN_ONLY, ptBlock.c:2271) has no matches, so several decoders
    can work inside the one stream — every piece but the first FINDS its block boundary (a complete dynamic header whose block decodes to
    text) and must end exactly where the next began.  Forced onto a small file with small pieces: the windows equal the one decoder's and
    the oracle loader's; a default-strategy file (matches) never starts the decoders; a stream that turns to matches half way falls back to
    the one decoder at a verified boundary; two members; a flipped bit and a truncated file fail loudly either way."""
    import gzip
    import zlib
    lengths = [2_511_003, 611_999, 40_004]
    hfile = str(tmp_path / "h.cov.gz")                              # Huffman-only (dense_cov writes it like the reference)
    synth.write_cov_dense(hfile, lengths, seed=3, min_run=20, max_run=120)
    text = gzip.open(hfile, "rb").read()
    assert len(text) > 600_000
    one, _ = _load_in_subprocess(hfile, {"HF_IO_PARALLEL": "0"})
    par, trace = _load_in_subprocess(hfile, SPEC)
    rounds, fallbacks = _rounds_and_fallbacks(trace)
    assert one.startswith("OK") and par == one and rounds >= 2 and fallbacks == 0, (one, par, trace)
    _same_store(fio.Table(hfile, 1_000_000, 4000).store(), _oracle_load_cov(hfile, 1_000_000, 4000, tmp_path))
    # the default strategy: matches from the first block on — the probe sees them, no decoder is started
    dfile = str(tmp_path / "d.cov.gz")
    with gzip.open(dfile, "wb", compresslevel=6) as f:
        f.write(text)
    got, trace = _load_in_subprocess(dfile, SPEC)
    assert got == one and _rounds_and_fallbacks(trace) == (0, 0), trace
    # literals only for the first 60 %, matches behind (one raw DEFLATE stream out of two: the first flushed, not finished)
    cut = text.rfind(b"\n", 0, int(len(text) * 0.6)) + 1
    c1 = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_HUFFMAN_ONLY)
    c2 = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY)
    raw = c1.compress(text[:cut]) + c1.flush(zlib.Z_FULL_FLUSH) + c2.compress(text[cut:]) + c2.flush()
    import struct
    mfile = str(tmp_path / "m.cov.gz")
    open(mfile, "wb").write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + raw + struct.pack("<II", zlib.crc32(text) & 0xffffffff, len(text) & 0xffffffff))
    assert gzip.open(mfile, "rb").read() == text
    got, trace = _load_in_subprocess(mfile, SPEC)
    rounds, fallbacks = _rounds_and_fallbacks(trace)
    assert got == one and rounds >= 1 and fallbacks == 1, (got, trace)
    # two members: the header lines and the first contig in one, the rest in another
    cut2 = text.find(b">hap_ctg1 ")
    tfile = str(tmp_path / "t.cov.gz")
    with open(tfile, "wb") as f:
        for part in (text[:cut2], text[cut2:]):
            c = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_HUFFMAN_ONLY)
            f.write(c.compress(part) + c.flush())
    got, trace = _load_in_subprocess(tfile, SPEC)
    assert got == one and _rounds_and_fallbacks(trace)[0] >= 2, (got, trace)
    # a flipped bit in the middle, a truncated file: an error with and without the parallel decoders
    blob = bytearray(open(hfile, "rb").read())
    blob[len(blob) // 2] ^= 0x10
    bad = str(tmp_path / "bad.cov.gz"); open(bad, "wb").write(bytes(blob))
    cutf = str(tmp_path / "cut.cov.gz"); open(cutf, "wb").write(open(hfile, "rb").read()[:-70_000])
    for path in (bad, cutf):
        for env in (SPEC, {"HF_IO_PARALLEL": "0"}):
            out, _ = _load_in_subprocess(path, env)
            assert out.startswith("ERROR"), (path, env, out)          # (the CRC-32 check, or — a changed digit — rows that no longer tile the contig)
