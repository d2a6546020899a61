#!/bin/bash
# A/B of two builds on the SAME GPU box (box-to-box variance is a few per cent): bash profiles/ab.sh "<EXTRA flags A>" "<EXTRA flags B>"
# rebuilds the library on the box with each EXTRA and runs the bench twice, alternating.  AB_BENCH_ARGS adds bench options
# (e.g. --no-kernel-events: no HIP events in the timed region, the per-kernel breakdown then comes from the extra passes).
set -u
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --steps 50 ${AB_BENCH_ARGS:-} | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_all']; print('$1', round(d['ms_per_step'],4), {a: round(b*1e3,1) for a,b in k.items()})"; }
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then X="$1"; else X="$2"; fi
    touch flagger_amd/csrc/hf_estep.hip
    make -C flagger_amd/csrc EXTRA="$X" > /dev/null 2>&1 || { echo "build failed for $v"; exit 1; }
    run "$v[$X]"
  done
done
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc > /dev/null 2>&1
