#!/usr/bin/env python
"""does the reciprocal normalisation of k_seg_fb's replays (f = nf * (1 / scale) instead of four divisions) contribute to the
last-printed-digit residue of `--accelerate` runs?  The twelve seeds of profiles/r04_squarem_residue.txt against the oracle command line, with
this round's library and with a build of the round-4 source with -DHF_SEG_RCP=0 (four divisions, as the reference: hmm.c:417,526) — built by
profiles/tools/build_variants.sh "norcp=-DHF_SEG_RCP=0" BEFORE the switch was removed from hf_seg.h, selected through LD_LIBRARY_PATH."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_cli as F  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NORCP = os.path.join(ROOT, "flagger_amd", "csrc", "variants", "norcp")
SEEDS = [8010, 8075, 8100, 8250, 8310, 8315, 8465, 8540, 8665, 8870, 8955, 8975]
WAYS = [("reciprocal (default)", {}), ("four divisions (HF_SEG_RCP=0)", {"LD_LIBRARY_PATH": NORCP + ":" + os.environ.get("LD_LIBRARY_PATH", "")})]
same = {w[0]: 0 for w in WAYS}
texts = {}
for seed in SEEDS:
    d, store, model, extra, args = F.make_case(seed, True)
    print(f"seed {seed} {model} {extra} windows {store.n_windows}")
    for k, (name, env) in enumerate(WAYS):
        outs = F.run_pair(d, args, (), env, tag=f"q{k}")
        if outs[0][0] != 0 or outs[1][0] != 0:
            print(f"   {name:32s} return codes {outs[0][0]} / {outs[1][0]}"); continue
        diff = F.compare(outs)
        same[name] += not diff
        texts[(seed, k)] = {n: open(os.path.join(outs[0][1], n)).read() for n in sorted(os.listdir(outs[0][1])) if n.endswith((".tsv", ".bed"))}
        print(f"   {name:32s} {'IDENTICAL to the oracle' if not diff else 'differs from the oracle in: ' + ', '.join(diff)}")
    if (seed, 0) in texts and (seed, 1) in texts:
        print("   the two builds' own outputs:", "identical" if texts[(seed, 0)] == texts[(seed, 1)] else
              "differ in " + ", ".join(n for n in texts[(seed, 0)] if texts[(seed, 0)][n] != texts[(seed, 1)].get(n)))
print("byte-identical to the oracle, of", len(SEEDS), "seeds:", same)
