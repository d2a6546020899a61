#!/usr/bin/env python
"""Round 6: the one plain-EM (no --accelerate) command line of 2 100 fuzzed ones whose files differ from the oracle's: what differs, and by how much."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "profiles", "tools"))
import fuzz_cli as F
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 30357
d, store, model, extra, args = F.make_case(seed, True)
print("seed", seed, model, extra, "windows", store.n_windows, "chunks", store.n_chunks, "regions", store.n_regions)
for tag, env in (("default", {}), ("two launches", {"HF_SEG_LAUNCHES": "2"}), ("per-chunk statistics", {"HF_STATS": "chunks"}), ("sequential kernels", None)):
    cli_args = ["--hipAlgo", "seq"] if env is None else []
    outs = F.run_pair(d, args, cli_args=cli_args, cli_env=env or {}, tag="p_" + tag.replace(" ", "_"))
    diff = F.compare(outs)
    print("[%s] differing files: %s" % (tag, diff))
    for f in diff:
        a = open(os.path.join(outs[0][1], f)).read().splitlines(); b = open(os.path.join(outs[1][1], f)).read().splitlines()
        n = 0
        for x, y in zip(a, b):
            if x != y and n < 4:
                print("     %s | hip: %s | oracle: %s" % (f, x[:160], y[:160])); n += 1
