"""CPU-only checks of the drop-in boundary: the shared library loads, exports every symbol the
headers declare, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from flagger_amd import _native as N
from flagger_amd import hmm, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(hf[msi]?o?_[a-z_0-9]+)\s*\(", txt))
    return {n for n in names if n not in ("hf_region_stride", "hf_stats_len")}  # static inline helpers


@pytest.mark.parametrize("header", ["hmm_flagger_hip.h", "hmm_flagger_model.h", "hmm_flagger_multi.h", "hmm_flagger_io.h",
                                    "hmm_flagger_summary.h"])
def test_every_declared_symbol_is_exported(header):
    L = C.CDLL(N.LIB_PATH)
    names = _declared(header)
    assert len(names) >= (15 if header != "hmm_flagger_summary.h" else 2)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_version_and_layout_helpers():
    L = N.lib()
    assert b"gfx950" in L.hf_version()
    assert N.stats_len(1, 10) == 1 + 24 * 10 + 16
    assert N.stats_len(7, 3) == 1 + 7 * (24 * 3 + 16)


@pytest.mark.skipif(N.lib().hf_device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    store = synth.config(1, scale=0.05)
    model = hmm.createModel(hmm.MODEL_TRUNC_EXP_GAUSSIAN, 3, store, np.zeros((4, 4)))
    with pytest.raises(N.HFError) as ei:
        hmm.EMList(store, model)
    assert ei.value.code == N.HF_E_NOGPU


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under flagger_amd/ may import, link or execute it, and the shared
    library must not depend on liboracle_hf.so."""
    import re
    import subprocess
    pkg = os.path.join(ROOT, "flagger_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+\S*oracle", text, re.M), f
                assert "liboracle" not in text and "ohf_" not in text and "oracle/" not in text, f
    so = os.path.join(pkg, "csrc", "libhmmflagger_hip.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in needed
