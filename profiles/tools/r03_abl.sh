#!/bin/bash
# ablation of the row fetch in the segment kernels (timing only, results are wrong): bash profiles/tools/r03_abl.sh "<EXTRA>" ...
set -u
cd "$(dirname "$0")/../.."
run() { python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_all']; print('$1', round(d['ms_per_step'],4), {a: round(b*1e3,1) for a,b in k.items()})"; }
for X in "$@"; do
  touch flagger_amd/csrc/hf_estep.hip
  make -C flagger_amd/csrc EXTRA="$X" > /dev/null 2>&1 || { echo "build failed $X"; continue; }
  run "[$X]"
done
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc > /dev/null 2>&1
