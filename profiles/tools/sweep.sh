#!/bin/bash
# Build-switch sweep on ONE GPU box: bash profiles/tools/sweep.sh "<EXTRA 1>" "<EXTRA 2>" ...   (each built and benched twice, interleaved)
set -u
cd "$(dirname "$0")/../.."
run() { python bench.py --no-cpu-baseline --steps 40 ${AB_BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_all']; print('$1', round(d['ms_per_step'],4), {a: round(b*1e3,1) for a,b in k.items()})"; }
for rep in 1 2; do
  for X in "$@"; do
    touch flagger_amd/csrc/hf_estep.hip
    make -C flagger_amd/csrc EXTRA="$X" > /dev/null 2>&1 || { echo "build failed for [$X]"; continue; }
    run "[$X]"
  done
done
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc > /dev/null 2>&1
