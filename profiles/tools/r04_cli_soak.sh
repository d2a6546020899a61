#!/bin/bash
# the command line, cold, N times on BASELINE configs[2] (.bin): every run must exit 0 and write the same bytes (the warm-up thread, the
# summary worker and the in-launch hand-offs are all concurrent machinery: a race would show as a crash or a differing file)
set -u
cd "$(dirname "$0")/../.."
N=${1:-150}
T=$(mktemp -d /tmp/clisoak.XXXX)
python - <<PY
import sys
sys.path.insert(0, ".")
from flagger_amd import synth
synth.config(2).write_bin("$T/cfg2.bin")
PY
bad=0; first=""
for i in $(seq 1 $N); do
  rm -rf $T/o; mkdir -p $T/o
  flagger_amd/csrc/hmm_flagger -i $T/cfg2.bin -n 100 -t 1e-3 -W 4000 -A tests/golden/alpha_hifi.tsv -w -o $T/o > $T/err 2>&1 || { echo "run $i: exit status $?"; tail -3 $T/err; bad=$((bad+1)); continue; }
  h=$(cat $T/o/*.tsv $T/o/*.bed | md5sum | cut -c1-32)
  if [ -z "$first" ]; then first=$h; fi
  if [ "$h" != "$first" ]; then echo "run $i: outputs differ ($h vs $first)"; bad=$((bad+1)); fi
done
echo "$N cold runs of the command line: $bad bad; $(ls $T/o | wc -l) files per run, md5 of all of them $first"
