"""The loader's own DEFLATE / gzip decoder (flagger_amd/csrc/hf_inflate.h, through hfio_gunzip) against zlib: every kind of block,
strategy, window and member layout zlib can produce, hand-made headers, and damaged streams (an error, never wrong bytes)."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from flagger_amd import _native as N


def _lib():
    L = N.lib()
    L.hfio_gunzip.restype = C.c_int
    L.hfio_gunzip.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_ubyte)), C.POINTER(C.c_size_t)]
    L.hfio_free.restype = None
    L.hfio_free.argtypes = [C.c_void_p]
    return L


def _gunzip(path):
    L = _lib()
    out = C.POINTER(C.c_ubyte)()
    n = C.c_size_t(0)
    rc = L.hfio_gunzip(str(path).encode(), C.byref(out), C.byref(n))
    if rc != 0:
        return rc, None
    data = C.string_at(out, n.value)
    L.hfio_free(out)
    return 0, data


def _gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8, wbits=15):
    c = zlib.compressobj(level, zlib.DEFLATED, 16 + wbits, mem, strategy)
    return c.compress(data) + c.flush()


def _payloads():
    rng = np.random.default_rng(7)
    rows = "".join(f"{i * 4000 + 1}\t{(i + 1) * 4000}\t{int(v)}\t{int(v)}\t0\t1\t0\t2\n" for i, v in enumerate(rng.normal(20, 5, 60_000).clip(0, 250)))
    return {
        "empty": b"",
        "one": b"x",
        "short": b"hello, hello, hello\n",
        "cov_rows": rows.encode(),                                          # what the loader reads: ~1.5 MB of text
        "zeros": bytes(300_000),                                            # distance 1, maximal lengths
        "period3": b"abc" * 100_000,                                        # distances below 8
        "period7": b"abcdefg" * 50_000,
        "random": rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes(),  # incompressible: stored blocks / long codes
        "skewed": rng.choice(np.arange(256, dtype=np.uint8), 400_000, p=np.r_[0.9, np.full(255, 0.1 / 255)]).tobytes(),   # 15-bit codes
        "far": (rng.integers(0, 256, 32_000, dtype=np.uint8).tobytes() + b"#") * 6,                                       # distances ~32 k
    }


@pytest.mark.parametrize("name", list(_payloads()))
def test_decoder_equals_zlib_for_every_encoder_setting(name, tmp_path):
    data = _payloads()[name]
    settings = [(lv, zlib.Z_DEFAULT_STRATEGY, 8, 15) for lv in (0, 1, 2, 4, 6, 9)]
    settings += [(6, zlib.Z_HUFFMAN_ONLY, 8, 15), (6, zlib.Z_RLE, 8, 15), (6, zlib.Z_FIXED, 8, 15), (6, zlib.Z_FILTERED, 8, 15),
                 (6, zlib.Z_DEFAULT_STRATEGY, 1, 15), (9, zlib.Z_DEFAULT_STRATEGY, 9, 15), (6, zlib.Z_DEFAULT_STRATEGY, 8, 9)]
    for k, (lv, st, mem, wb) in enumerate(settings):
        p = tmp_path / f"{name}_{k}.gz"
        p.write_bytes(_gz(data, lv, st, mem, wb))
        rc, got = _gunzip(p)
        assert rc == 0 and got == data, (name, lv, st, mem, wb, rc)


def test_members_headers_and_trailing_bytes(tmp_path):
    a, b = b"first member\n" * 1000, b"second member\n" * 3000
    p = tmp_path / "two.gz"
    p.write_bytes(_gz(a) + _gz(b, 9))                                       # concatenated gzip files are one stream (gzread reads both)
    assert _gunzip(p) == (0, a + b)
    p = tmp_path / "garbage_after.gz"
    p.write_bytes(_gz(a) + b"\x00\x00not a member")                        # ... and bytes that are no member are ignored
    assert _gunzip(p) == (0, a)
    # FEXTRA + FNAME + FCOMMENT + FHCRC by hand around a raw deflate stream
    raw = zlib.compress(a, 6)[2:-4]
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\0\0\0\0\0\x03" + struct.pack("<H", 5) + b"extra" + b"name.cov\0" + b"a comment\0"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xffff)
    p = tmp_path / "flags.gz"
    p.write_bytes(hdr + raw + struct.pack("<II", zlib.crc32(a), len(a)))
    assert _gunzip(p) == (0, a)
    with gzip.open(tmp_path / "py.gz", "wb") as f:                          # python's writer: FNAME, mtime
        f.write(b)
    assert _gunzip(tmp_path / "py.gz") == (0, b)
    (tmp_path / "plain.txt").write_bytes(a)
    assert _gunzip(tmp_path / "plain.txt")[0] == -3


def test_damaged_streams_are_errors_never_wrong_bytes(tmp_path):
    data = _payloads()["cov_rows"][:200_000]
    for lv, st in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED)):
        z = _gz(data, lv, st)
        p = tmp_path / "cut.gz"
        for cut in list(range(0, 40)) + list(range(40, len(z), max(1, len(z) // 97))) + [len(z) - k for k in range(1, 10)]:
            p.write_bytes(z[:cut])
            rc, got = _gunzip(p)
            assert rc != 0 and got is None, (lv, st, cut, rc)
        rng = np.random.default_rng(3)
        for _ in range(300):                                                # one flipped bit anywhere: an error, or (a bit of the header's
            k = int(rng.integers(0, len(z)))                                # mtime / OS fields) the same bytes
            bad = bytearray(z)
            bad[k] ^= 1 << int(rng.integers(0, 8))
            p.write_bytes(bytes(bad))
            rc, got = _gunzip(p)
            assert (rc != 0 and got is None) or got == data, (lv, st, k, rc)


def test_loader_reads_concatenated_gzip_members_like_gzread(tmp_path):
    from flagger_amd import io as fio, synth
    st = synth.synthesize([400_000, 90_000], 1000, 100_000, [20], seed=4)
    st.write_cov(str(tmp_path / "one.cov"))
    text = (tmp_path / "one.cov").read_bytes()
    cut = text.index(b"\n", len(text) // 2) + 1 + 7                         # the second member starts in the middle of a line
    (tmp_path / "two.cov.gz").write_bytes(_gz(text[:cut]) + _gz(text[cut:], 1))
    a = fio.Table(str(tmp_path / "one.cov"), 100_000, 1000).store()
    b = fio.Table(str(tmp_path / "two.cov.gz"), 100_000, 1000).store()
    for name in ("cov", "mapq", "clip", "annot", "chunk_off", "chunk_s", "chunk_e"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    os.environ["HF_IO_ZLIB"] = "1"                                          # the zlib path of the same reader
    try:
        c = fio.Table(str(tmp_path / "two.cov.gz"), 100_000, 1000).store()
    finally:
        del os.environ["HF_IO_ZLIB"]
    assert np.array_equal(a.cov, c.cov) and np.array_equal(a.chunk_off, c.chunk_off)


def test_a_file_cut_inside_the_next_members_header_is_an_error(tmp_path):
    """after a complete member, bytes that START a gzip member (1f 8b) but end inside its header were taken for the
    end of the file — a concatenated / bgzip input cut there loaded silently without its last rows."""
    from flagger_amd import io as fio, synth
    a, b = b"first member\n" * 1000, b"second member\n" * 3000
    za, zb = _gz(a), _gz(b)
    p = tmp_path / "cut_header.gz"
    for keep in (2, 3, 5, 9):                                               # magic (+ part of the fixed 10 bytes)
        p.write_bytes(za + zb[:keep])
        rc, got = _gunzip(p)
        assert rc == -2 and got is None, (keep, rc)
    hdr = b"\x1f\x8b\x08" + bytes([4]) + b"\0\0\0\0\0\x03" + struct.pack("<H", 50) + b"short"   # FEXTRA longer than the file
    p.write_bytes(za + hdr)
    assert _gunzip(p)[0] == -2
    p.write_bytes(za + b"\x1f")                                             # one stray byte is no member: ignored
    assert _gunzip(p) == (0, a)
    # the loader itself (mapped file, threaded reader): an error, not a shorter table
    st = synth.synthesize([400_000, 90_000], 1000, 100_000, [20], seed=4)
    st.write_cov(str(tmp_path / "one.cov"))
    text = (tmp_path / "one.cov").read_bytes()
    cut = text.index(b"\n", len(text) // 2) + 1
    z2 = _gz(text[cut:], 1)
    (tmp_path / "cut.cov.gz").write_bytes(_gz(text[:cut]) + z2[:6])
    with pytest.raises(Exception):
        fio.Table(str(tmp_path / "cut.cov.gz"), 100_000, 1000)
