#!/usr/bin/env python
"""the loader (N1: chunk.c:393-483, track_reader.c:751-818) at the row density of a real bam2cov track — BASELINE
configs[2]'s genome (2 x 3.03 Gb) with coverage / mapq / clip changing every 20-220 bases: ~51 M rows, ~1.5 GB of text, ONE DEFLATE stream.
Reports rows/s, inflated MB/s, peak RSS; proves the windows of a sample of contigs against the oracle's per-base loader (each sampled contig
written alone by the same generator: the same rows it has in the full file); then the command line end to end with its phase times.
    python profiles/tools/r06_loader_dense.py [scale]"""
import ctypes as C
import os
import resource
import subprocess
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from flagger_amd import synth, io as fio  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
tmp = os.environ.get("DENSE_TMP", "/tmp/dense")
os.makedirs(tmp, exist_ok=True)
L = synth.human_diploid_lengths(scale)
full = os.path.join(tmp, "dense_full.cov.gz")
t0 = time.perf_counter()
info = synth.write_cov_dense(full, L, seed=5, min_run=20, max_run=220)
print("generated %s: %d rows, %d bases, %.2f GB of text, %.0f MB compressed, %.0f s" %
      (full, info["rows"], info["bases"], info["text_bytes"] / 1e9, os.path.getsize(full) / 1e6, time.perf_counter() - t0))
for env, label in (({}, "own DEFLATE decoder (default)"), ({"HF_IO_ZLIB": "1"}, "zlib gzread (HF_IO_ZLIB=1)")):
    code = ("import sys,time,resource\nsys.path.insert(0,%r)\nfrom flagger_amd import io as fio\nt=time.perf_counter()\n"
            "tab=fio.Table(%r, 20000000, 4000)\ndt=time.perf_counter()-t\n"
            "print(tab._L.hfio_n_windows(tab._h), tab._L.hfio_n_chunks(tab._h), dt, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss)\n") % (ROOT, full)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, HF_IO_TRACE="1", **env))
    nw, nc, dt, rss = r.stdout.split()
    print("load [%s]: %s windows, %s chunks in %.2f s = %.1f M rows/s, %.0f MB/s of text; peak RSS %.0f MB | %s" %
          (label, nw, nc, float(dt), info["rows"] / float(dt) / 1e6, info["text_bytes"] / float(dt) / 1e6, float(rss) / 1024, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else ""))
# the windows of a sample of contigs against the oracle's per-base loader
tab = fio.Table(full, 20_000_000, 4000)
st = tab.store()
sample = [20, 21, len(L) // 2 + 19] if len(L) > 40 else [1]
exe = os.path.join(ROOT, "oracle", "hf_oracle")
bases = 0
for ci in sample:
    one = os.path.join(tmp, "c%d.cov.gz" % ci)
    synth.write_cov_dense(one, L, seed=5, min_run=20, max_run=220, only=ci)
    out = os.path.join(tmp, "o%d" % ci); os.makedirs(out, exist_ok=True)
    t1 = time.perf_counter()
    subprocess.run([exe, "-i", one, "-o", out, "-C", "20000000", "-W", "4000", "-n", "0", "-B", "-p", "2", "-e"], check=True, capture_output=True)
    ref = synth.WindowStore.read_bin(os.path.join(out, "chunks.c_20000000.w_4000.bin"))
    sel = [c for c in range(st.n_chunks) if st.chunk_ctg[c] == "hap_ctg%d" % ci]
    mine = st.subset_chunks(sel)
    same = all(np.array_equal(getattr(mine, f), getattr(ref, f)) for f in ("cov", "mapq", "clip", "annot", "chunk_off", "chunk_s", "chunk_e", "chunk_ctg_len"))
    bases += L[ci]
    print("contig %d (%d bases, %d windows): product loader's windows out of the full file == oracle loader's of the contig alone: %s (oracle %.1f s)" %
          (ci, L[ci], mine.n_windows, same, time.perf_counter() - t1))
    assert same
print("sampled %.1f %% of the bases" % (100.0 * bases / sum(L)))
# the command line end to end (needs a GPU)
if os.environ.get("DENSE_CLI", "1") == "1":
    cli = os.path.join(ROOT, "flagger_amd", "csrc", "hmm_flagger")
    out = os.path.join(tmp, "cli"); os.makedirs(out, exist_ok=True)
    cmd = [cli, "-i", full, "-W", "4000", "-A", os.path.join(ROOT, "tests", "golden", "alpha_hifi.tsv"), "-n", "100", "-t", "1e-3", "-o", out]
    t1 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, HF_CLI_TIMING="1"))
    print("hmm_flagger end to end: rc %d, %.2f s wall" % (r.returncode, time.perf_counter() - t1))
    for line in r.stderr.splitlines():
        if "[phase]" in line or "Peak RSS" in line or "EM+decode" in line or "chunks are parsed" in line or "converged" in line:
            print("   " + line.strip())
    bed = os.path.join(out, "final_flagger_prediction.bed")
    if os.path.exists(bed):
        import collections
        cnt = collections.Counter(l.split("\t")[3] for l in open(bed) if not l.startswith("track"))
        print("   final BED: %d intervals %s" % (sum(cnt.values()), dict(cnt)))
