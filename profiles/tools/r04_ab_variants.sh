#!/bin/bash
# same-box comparison of prebuilt library variants (profiles/tools/build_variants.sh) by rocprofv3's own kernel durations:
#   [BENCH_ARGS=..] [ROUNDS=2] bash profiles/tools/r04_ab_variants.sh old new noRCP ...
set -u
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp
for rep in $(seq 1 ${ROUNDS:-2}); do
for X in "$@"; do
  ( cd /tmp && rm -rf /tmp/ktab && HF_LIBRARY_VARIANT=$X rocprofv3 --kernel-trace --stats -d /tmp/ktab -o kt -- python $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-kernel-events --no-em-run ${BENCH_ARGS:-} > /tmp/ab.json 2> /tmp/ab.err )
  DB=$(find /tmp/ktab -name '*_results.db' | head -1)
  python $R/profiles/summarize.py $DB /tmp/ab.csv > /dev/null
  python - <<PY
import csv, json
try:
    ms = json.loads([l for l in open("/tmp/ab.json") if l.startswith("{")][-1])["ms_per_step"]
    rows = {r["kernel"].split("(")[0].replace("void ", ""): float(r["avg_us"]) for r in csv.DictReader(open("/tmp/ab.csv")) if int(r["calls"]) > 50}
    print("%-10s" % "$X", "step (under rocprof) %.4f ms" % ms, {k[:22]: round(v, 2) for k, v in rows.items() if k.startswith("k_")})
except Exception as e:
    print("$X", "failed:", e, open("/tmp/ab.err").read()[-500:])
PY
done
done
