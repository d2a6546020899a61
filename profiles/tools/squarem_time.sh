#!/bin/bash
# N2: the command line with --accelerate (SQUAREM) against plain EM on BASELINE configs[2] (.bin input):
# passes, EM+decode time and iterations to convergence.  Run on a GPU box from the repo root.
set -u
T=$(mktemp -d /tmp/sqm.XXXX)
python - <<PY
import sys
sys.path.insert(0, ".")
from flagger_amd import synth
synth.config(2).write_bin("$T/cfg2.bin")
PY
for mode in "" "--accelerate"; do
  mkdir -p $T/o$mode
  flagger_amd/csrc/hmm_flagger -i $T/cfg2.bin -n 100 -W 4000 -A tests/golden/alpha_hifi.tsv $mode -o $T/o$mode > $T/err 2>&1
  echo "[$mode] rc=$? $(grep -o 'Parameters converged after [0-9]* iterations\|Parameter estimation stopped.*after [0-9]* iterations' $T/err)  $(grep -o 'EM+decode: [0-9]* passes over [0-9]* windows in [0-9.]* s' $T/err)  final LL $(tail -1 $T/o$mode/loglikelihood.tsv | cut -f3)"
done
