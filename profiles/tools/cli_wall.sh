#!/bin/bash
# Wall time of the whole command line on BASELINE configs[2] read from a .cov.gz (one run per window), 100 EM iterations
# (default convergence test), all outputs written.  Run on a GPU box from the repo root.
set -u
T=$(mktemp -d /tmp/cliwall.XXXX)
python - <<PY
import sys
sys.path.insert(0, ".")
from flagger_amd import synth
synth.config(2).write_cov("$T/cfg2.cov.gz")
PY
mkdir -p $T/o
for rep in 1 2 3; do
  S=$(date +%s%N)
  HF_CLI_TIMING=1 flagger_amd/csrc/hmm_flagger -i $T/cfg2.cov.gz -n 100 -W 4000 -A tests/golden/alpha_hifi.tsv -o $T/o > $T/err 2>&1
  grep "^\[phase\]" $T/err | cut -c1-400; echo "rc=$? wall $(( ($(date +%s%N) - S) / 1000000 )) ms   $(grep -o 'EM+decode: [0-9]* passes.*' $T/err | cut -c1-200)"
done
grep -c . $T/o/final_flagger_prediction.bed
