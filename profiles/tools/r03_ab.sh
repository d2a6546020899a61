#!/bin/bash
# same-box comparison of build switches or environments: bash profiles/tools/r03_ab.sh "<EXTRA flags or ENV=..>" ...   (an argument that
# contains '=' and no '-D' is taken as environment for bench.py, anything else as make EXTRA)
set -u
cd "$(dirname "$0")/../.."
run() { env $2 python bench.py --no-cpu-baseline --steps 40 ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_all']; print('$1', round(d['ms_per_step'],4), {a: round(b*1e3,1) for a,b in k.items()})"; }
for X in "$@"; do
  if [[ "$X" == *=* && "$X" != *-D* ]]; then run "[$X]" "$X"; continue; fi
  touch flagger_amd/csrc/hf_estep.hip
  make -C flagger_amd/csrc EXTRA="$X" > /dev/null 2>&1 || { echo "build failed $X"; continue; }
  run "[$X]" "A=1"
done
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc > /dev/null 2>&1
