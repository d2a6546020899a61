#!/usr/bin/env python
"""attribute the last-printed-digit residue of `--accelerate` runs.  The 12 seeds of profiles/r03_fuzz.txt whose
outputs differed from the oracle command line (FUZZ_OPTIONS=1 axis), each re-run four ways:
   default            statistics by emission row, segment scan kernels (f.(T.e), fused sums)
   HF_STATS=chunks    per-chunk statistics summed in chunk-list order (the reference's merge order), scan kernels
   --hipAlgo seq      sequential forward / backward in the reference's operation order (per-chunk statistics)
   seq + HF_STATS=chunks   (the same as seq: HF_ALGO_SEQ always takes the per-chunk statistics; kept to show it)
and which of them is byte-identical to the oracle.  python profiles/tools/r04_residue.py [seed ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_cli as F  # noqa: E402

SEEDS = [8010, 8075, 8100, 8250, 8310, 8315, 8465, 8540, 8665, 8870, 8955, 8975]
WAYS = [("default", (), {}), ("HF_STATS=chunks", (), {"HF_STATS": "chunks"}), ("--hipAlgo seq", ("--hipAlgo", "seq"), {}),
        ("seq + chunks", ("--hipAlgo", "seq"), {"HF_STATS": "chunks"})]
seeds = [int(a) for a in sys.argv[1:]] or SEEDS
closed = {w[0]: 0 for w in WAYS}
for seed in seeds:
    d, store, model, extra, args = F.make_case(seed, True)
    print(f"seed {seed} {model} {extra} windows {store.n_windows}")
    for k, (name, cargs, env) in enumerate(WAYS):
        outs = F.run_pair(d, args, cargs, env, tag=f"p{k}")
        if outs[0][0] != 0 or outs[1][0] != 0:
            print(f"   {name:18s} return codes {outs[0][0]} / {outs[1][0]}"); continue
        diff = F.compare(outs)
        closed[name] += not diff
        print(f"   {name:18s} {'IDENTICAL' if not diff else 'differs: ' + ', '.join(diff)}")
        if diff and os.environ.get("FUZZ_SHOW"):
            F.show(outs, diff, 3)
print("byte-identical to the oracle, of", len(seeds), "seeds:", closed)
