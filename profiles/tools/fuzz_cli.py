#!/usr/bin/env python
"""One-off fuzz of the drop-in claim: random small inputs through the product command line and the oracle command line
(-n 15, outputs compared byte for byte).  python profiles/tools/fuzz_cli.py <first seed> <n seeds>"""
import filecmp
import os
import subprocess
import sys
import tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from flagger_amd import synth  # noqa: E402

CLI = os.path.join(ROOT, "flagger_amd", "csrc", "hmm_flagger"); ORC = os.path.join(ROOT, "oracle", "hf_oracle")
ALPHA = os.path.join(ROOT, "tests", "golden", "alpha_hifi.tsv")
FILES = ["final_flagger_prediction.bed", "loglikelihood.tsv", "emission_final.tsv", "transition_final.tsv"]


def make_case(seed, with_options):
    """The seeded random input of a run (written as in.bin into a fresh directory) and its command-line arguments."""
    rng = np.random.default_rng(9000 + seed)
    window_len = int(rng.choice([1000, 4000]))
    lengths = [int(rng.integers(50, 3000)) * window_len + int(rng.integers(0, window_len)) for _ in range(int(rng.integers(1, 5)))]
    R = int(rng.integers(1, 4))
    store = synth.synthesize(lengths, window_len, int(rng.choice([50, 300])) * window_len, [int(rng.integers(10, 40)) for _ in range(R)],
                             seed=seed, avg_alignment_len=int(rng.choice([0, 15_000])), region_run_bases=(5 * window_len, 300 * window_len))
    d = tempfile.mkdtemp()
    store.write_bin(os.path.join(d, "in.bin"))
    model = ["trunc_exp_gaussian", "gaussian", "negative_binomial"][seed % 3]
    extra = ["--accelerate"] if seed % 5 == 0 else []
    if with_options:           # second axis: the options that change the E-step's inputs
        if rng.random() < 0.25: extra += ["-e"]
        if rng.random() < 0.3: extra += ["-q", "%.2f" % rng.uniform(0.1, 0.9), "--minHighMapqRatio", "%.2f" % rng.uniform(0.1, 0.9)]   # the reference has no short -Q
        if rng.random() < 0.3: extra += ["-p", str(int(rng.integers(2, 9)))]
        if rng.random() < 0.3: extra += ["-f", "%.2f" % rng.uniform(0.5, 1.0)]
        if rng.random() < 0.3: extra += ["-P"]
        if rng.random() < 0.2: extra += ["-M", "%d,%d,%d" % tuple(int(x) * window_len for x in rng.integers(1, 6, 3))]
    args = ["-i", os.path.join(d, "in.bin"), "-n", "15", "-W", str(window_len), "-m", model] + ([] if model == "negative_binomial" else ["-A", ALPHA]) + extra
    return d, store, model, extra, args


def run_pair(d, args, cli_args=(), cli_env=None, tag="p"):
    """Both command lines on the case; returns (return codes, output directories)."""
    outs = []
    for exe, name in ((CLI, tag), (ORC, "o")):
        o = os.path.join(d, name)
        if exe == ORC and os.path.exists(o):
            outs.append((0, o)); continue
        os.mkdir(o)
        r = subprocess.run([exe] + args + ["-o", o] + (["--threads", "8"] if exe == ORC else list(cli_args)), capture_output=True, text=True,
                           env=dict(os.environ, **(cli_env or {})) if exe == CLI else None)
        outs.append((r.returncode, o))
    return outs


def compare(outs):
    names = FILES + [f for f in sorted(os.listdir(outs[1][1])) if f not in FILES and ("posterior" in f or f.startswith("prediction_summary"))]
    return [f for f in names if not (os.path.exists(os.path.join(outs[0][1], f)) and
                                     filecmp.cmp(os.path.join(outs[0][1], f), os.path.join(outs[1][1], f), shallow=False))]


def show(outs, diff, limit=6):
    for f in diff:
        a = open(os.path.join(outs[0][1], f)).read().splitlines(); b = open(os.path.join(outs[1][1], f)).read().splitlines()
        shown = 0
        for x, y in zip(a, b):
            if x != y and shown < limit:
                print("   ", f, "|", x[:150], "|", y[:150]); shown += 1


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    bad = 0
    n_acc = 0
    for seed in range(first, first + count):
        if os.environ.get("FUZZ_ONLY_ACCELERATED") and seed % 5 != 0:     # (round 6: the --accelerate runs only — every fifth seed)
            continue
        n_acc += seed % 5 == 0
        d, store, model, extra, args = make_case(seed, bool(os.environ.get("FUZZ_OPTIONS")))
        outs = run_pair(d, args)
        if outs[0][0] != outs[1][0]:
            bad += 1; print("seed", seed, model, extra, "return codes differ", outs[0][0], outs[1][0]); continue
        if outs[0][0] != 0:
            continue
        diff = compare(outs)
        if diff:
            bad += 1; print("seed", seed, model, extra, "windows", store.n_windows, "DIFFERENT:", diff)
            if os.environ.get("FUZZ_SHOW"):          # the lines that differ (first 6 per file)
                show(outs, diff)
    print("seeds", first, "..", first + count - 1, "runs with a difference:", bad, "(accelerated runs among them: %d)" % n_acc)
