#!/usr/bin/env python
"""N1: time of the .cov.gz -> window table step on BASELINE configs[2] written as one run per window
(1.53 M rows, 6.06 Gb): the product's run-length-aware loader (hfio_load) against the oracle's per-base restatement
of the reference loader (chunk.c:393-547; on a 2 % sample, it is O(bases)).  Host-only, no GPU."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from flagger_amd import synth  # noqa: E402
from flagger_amd.io import Table  # noqa: E402

tmp = tempfile.mkdtemp()
full = synth.config(2)
p_full = os.path.join(tmp, "cfg2.cov.gz")
full.write_cov(p_full)
t = time.perf_counter()
tb = Table(p_full, 20_000_000, 4000)
dt = time.perf_counter() - t
st = tb.store()
assert np.array_equal(st.cov, full.cov) and np.array_equal(st.annot, full.annot) and np.array_equal(st.chunk_off, full.chunk_off)
bases = int(sum(int(full.chunk_e[c]) - int(full.chunk_s[c]) + 1 for c in range(full.n_chunks)))
print(f"product loader: {full.n_windows} rows, {bases / 1e9:.2f} Gb, {os.path.getsize(p_full) / 1e6:.1f} MB gz in {dt:.2f} s "
      f"= {full.n_windows / dt / 1e6:.2f} M rows/s")

small = synth.config(2, scale=0.02)
p_small = os.path.join(tmp, "cfg2_small.cov.gz")
small.write_cov(p_small)
out = os.path.join(tmp, "o")
os.mkdir(out)
t = time.perf_counter()
subprocess.run([os.path.join(ROOT, "oracle", "hf_oracle"), "-i", p_small, "-n", "0", "-W", "4000", "-o", out, "--dumpBin"],
               check=True, capture_output=True)
dto = time.perf_counter() - t
t = time.perf_counter()
Table(p_small, 20_000_000, 4000)
dts = time.perf_counter() - t
sb = int(sum(int(small.chunk_e[c]) - int(small.chunk_s[c]) + 1 for c in range(small.n_chunks)))
print(f"2 % sample ({sb / 1e6:.0f} Mb): oracle CLI (per-base loader + one pass + outputs) {dto:.2f} s, product loader {dts:.3f} s "
      f"=> x{dto / dts:.0f}; per-base loader extrapolated to the full input: {dto * bases / sb:.0f} s")
