#!/bin/bash
# same-box comparison of build switches by rocprofv3's own kernel durations (HIP events around a kernel are +-3 us):
#   [BENCH_ARGS=..] bash profiles/tools/r03_ab_rocprof.sh "<EXTRA flags>" ...
set -u
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp
for X in "$@"; do
  touch flagger_amd/csrc/hf_estep.hip
  make -C flagger_amd/csrc EXTRA="$X" > /dev/null 2>&1 || { echo "build failed $X"; continue; }
  ( cd /tmp && rm -rf /tmp/ktab && rocprofv3 --kernel-trace --stats -d /tmp/ktab -o kt -- python $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-kernel-events ${BENCH_ARGS:-} > /tmp/ab.json 2> /dev/null )
  DB=$(find /tmp/ktab -name '*_results.db' | head -1)
  python $R/profiles/summarize.py $DB /tmp/ab.csv > /dev/null
  python - <<PY
import csv, json
ms = json.loads([l for l in open("/tmp/ab.json") if l.startswith("{")][-1])["ms_per_step"]
rows = {r["kernel"].split("(")[0].replace("void ", ""): float(r["avg_us"]) for r in csv.DictReader(open("/tmp/ab.csv")) if int(r["calls"]) > 50}
print("[$X]", "step (under rocprof) %.4f ms" % ms, {k[:22]: round(v, 2) for k, v in rows.items() if k.startswith("k_")})
PY
done
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc > /dev/null 2>&1
