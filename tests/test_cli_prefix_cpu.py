"""getopt_long accepts any unambiguous prefix of a long option, and the reference's WDL relies on it (`--alpha` for
`--alphaTsv`, hmm_flagger.wdl:52).  The drop-in command line adds four options of its own (--device, --hipAlgo, --gpus,
--exchange): every prefix that was unique among the reference's options (hmm_flagger.c:578-608) must still be unique and
resolve to the same option.  Checked on the command line itself (no GPU needed: option errors come before any device use)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "flagger_amd", "csrc", "hmm_flagger")

# the reference's long options (hmm_flagger.c:578-608): name -> takes an argument
REFERENCE = {"input": 1, "preset": 1, "iterations": 1, "convergenceTol": 1, "disableAdjustContigEnds": 0, "minReadFractionAtEnds": 1,
             "modelType": 1, "maxHighMapqRatio": 1, "minHighMapqRatio": 1, "chunkLen": 1, "windowLen": 1, "contigsList": 1, "threads": 1,
             "collapsedComps": 1, "alphaTsv": 1, "binArrayFile": 1, "writeParameterStatsPerIteration": 0,
             "writeBenchmarkingStatsPerIteration": 0, "writePosteriorProbs": 0, "outputDir": 1, "overlapRatioThreshold": 1,
             "labelNames": 1, "initialRandomDev": 1, "trackName": 1, "dumpBin": 0, "accelerate": 0, "minimumLengths": 1}
ADDED = ["device", "hipAlgo", "gpus", "exchange"]


def unique_prefixes(names):
    out = []
    for n in names:
        for k in range(1, len(n) + 1):
            p = n[:k]
            if sum(1 for m in names if m.startswith(p)) == 1 or p == n:
                out.append((p, n))
    return out


def test_added_options_shadow_no_prefix_of_a_reference_option():
    ref = unique_prefixes(list(REFERENCE))
    now = dict(unique_prefixes(list(REFERENCE) + ADDED))
    lost = [(p, n) for p, n in ref if now.get(p) != n]
    assert not lost, lost


@pytest.mark.skipif(not os.path.exists(CLI), reason="hmm_flagger not built")
def test_the_binary_resolves_them():
    """One process per option with its SHORTEST unique prefix (and `--alpha`, the WDL's spelling): getopt must not call it
    ambiguous or unknown.  The command then stops at its own argument checks (no input / no such file)."""
    ref = unique_prefixes(list(REFERENCE))
    shortest = {}
    for p, n in ref:
        if n not in shortest or len(p) < len(shortest[n]):
            shortest[n] = p
    shortest["alphaTsv+wdl"] = "alpha"
    for n, p in shortest.items():
        name = n.split("+")[0]
        args = [CLI, "--" + p] + (["1"] if REFERENCE[name] else [])
        r = subprocess.run(args, capture_output=True, text=True)
        assert "ambiguous" not in r.stderr and "unrecognized" not in r.stderr and "undefined option" not in r.stderr, (p, r.stderr[-300:])
