"""The binding INTEGRATION.md documents, compiled for real: integration/hmm_hip_shim.c — the file a maintainer of
mobinasri/flagger adds next to hmm.c — builds against tests/shim_mock/hmm.h (the reference's struct fields) with
-Wall -Wextra -Werror, links with libhmmflagger_hip.so and, without a GPU, fails the way the reference's own fatal errors do
(message on stderr, EXIT_FAILURE) instead of computing anything on the CPU.  tests/test_shim_gpu.py runs it on a GPU."""
import os
import struct
import subprocess

import numpy as np
import pytest

from flagger_amd import _native as N
from flagger_amd import hmm, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flagger_amd", "csrc")


def build_driver(tmp_path):
    exe = str(tmp_path / "shim_driver")
    ORC = os.path.join(ROOT, "oracle")     # the mock's NegativeBinomial functions sit on the oracle (test infrastructure only)
    cmd = ["gcc", "-std=gnu11", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tests", "shim_mock"),
           "-I" + os.path.join(ROOT, "include"), "-I" + ORC, os.path.join(ROOT, "integration", "hmm_hip_shim.c"),
           os.path.join(ROOT, "tests", "shim_mock", "driver.c"), os.path.join(ROOT, "tests", "shim_mock", "mock_nb.c"),
           "-L" + CSRC, "-lhmmflagger_hip", "-Wl,-rpath," + CSRC, "-L" + ORC, "-loracle_hf", "-Wl,-rpath," + ORC, "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def write_dump(path, store, model, adjust, min_frac):
    R, K = model.numberOfRegions, model.maxNumberOfComps
    p = model.params()
    with open(path, "wb") as f:
        f.write(struct.pack("<7i", store.n_chunks, R, K, model.modelType, store.window_len, store.avg_alignment_len, int(adjust)))
        L = N.lib()
        f.write(struct.pack("<4d", min_frac, L.hfm_max_high_mapq_ratio(model._h), L.hfm_min_high_mapq_ratio(model._h),
                            L.hfm_min_highly_clipped_ratio(model._h)))
        f.write(np.asarray([[p.alpha[i][j] for j in range(4)] for i in range(4)], dtype="<f8").tobytes())
        for c in range(store.n_chunks):
            f.write(struct.pack("<4i", int(store.chunk_off[c + 1] - store.chunk_off[c]), int(store.chunk_s[c]), int(store.chunk_e[c]),
                                int(store.chunk_ctg_len[c])))
        for a, dt in ((store.cov, "<u2"), (store.mapq, "<u2"), (store.clip, "<u2"), (store.annot, "<u8")):
            f.write(np.asarray(a).astype(dt).tobytes())
        f.write(model.param_vector().astype("<f8").tobytes())


def test_the_documented_shim_compiles_links_and_has_no_cpu_fallback(tmp_path):
    exe = build_driver(tmp_path)
    store = synth.config(2, scale=0.004)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 4, store, synth.HIFI_ALPHA)
    dump = str(tmp_path / "in.bin")
    write_dump(dump, store, model, True, 0.95)
    r = subprocess.run([exe, dump, str(tmp_path / "out.bin")], capture_output=True, text=True)
    if N.lib().hf_device_count() == 0:
        assert r.returncode == 1 and "no HIP device" in r.stderr and not os.path.exists(str(tmp_path / "out.bin"))
    else:
        assert r.returncode == 0, r.stderr


def test_integration_md_prints_the_compiled_shim():
    """INTEGRATION.md section 1 is what a maintainer copies from: its listing must be the file the tests compile, byte for byte
    (the document still showed the model-type cast that round 2 had fixed in the file)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shim = open(os.path.join(ROOT, "integration", "hmm_hip_shim.c")).read()
    start = doc.index("```c\n/* hmm_hip_shim.c") + len("```c\n")
    end = doc.index("\n```\n", start)
    assert doc[start:end] == shim.rstrip("\n"), "INTEGRATION.md section 1 differs from integration/hmm_hip_shim.c: regenerate the listing"
    assert "? HF_MODEL_GAUSSIAN : HF_MODEL_TRUNC_EXP_GAUSSIAN" not in doc
