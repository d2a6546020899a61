#!/usr/bin/env python
"""One-launch segment kernel at 8x the BASELINE workload (24 k workgroups, eight dispatch rounds): same bits as two launches, no time-out."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_estep_gpu import _pass_in_subprocess
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
a = _pass_in_subprocess({"HF_SEG_LAUNCHES": "1"}, scale, passes=3)
b = _pass_in_subprocess({"HF_SEG_LAUNCHES": "2"}, scale, passes=3)
print("scale", scale, "fallback" if "falls back" in a[3] else "no time-out", "| identical:", a[0] == b[0] and a[2] == b[2] and bool(np.array_equal(a[1], b[1])), "| log-likelihoods", a[0])
