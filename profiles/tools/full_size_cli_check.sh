#!/bin/bash
# End-to-end drop-in check at BASELINE configs[2] size: the command line of this build and the oracle command line on the
# same .bin (1.53 M windows), 10 EM iterations: every output file must be identical.  Run on a GPU box from the repo root.
set -u
T=$(mktemp -d /tmp/fullcli.XXXX)
python - <<PY
import sys
sys.path.insert(0, ".")
from flagger_amd import synth
synth.config(2).write_bin("$T/cfg2.bin")
PY
mkdir -p $T/gpu $T/cpu
ARGS="-i $T/cfg2.bin -n 10 -W 4000 -A tests/golden/alpha_hifi.tsv -w"
S=$(date +%s%N); flagger_amd/csrc/hmm_flagger $ARGS -o $T/gpu > $T/gpu.err 2>&1; echo "gpu cli rc=$? wall $(( ($(date +%s%N) - S) / 1000000 )) ms"; tail -3 $T/gpu.err | cut -c1-200
S=$(date +%s%N); oracle/hf_oracle $ARGS --threads 16 -o $T/cpu > $T/cpu.err 2>&1; echo "oracle cli rc=$? wall $(( ($(date +%s%N) - S) / 1000000 )) ms"; tail -2 $T/cpu.err | cut -c1-200
for f in final_flagger_prediction.bed loglikelihood.tsv emission_final.tsv transition_final.tsv emission_iteration_5.tsv; do
  cmp -s $T/gpu/$f $T/cpu/$f && echo "identical: $f ($(wc -l < $T/gpu/$f) lines)" || echo "DIFFERENT: $f"
done
ls $T/gpu | tr '\n' ' '
