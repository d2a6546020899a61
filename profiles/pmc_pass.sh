#!/bin/bash
# One rocprofv3 --pmc pass over a short bench run; prints the per-kernel averages.  bash profiles/pmc_pass.sh <out.json> COUNTER...
set -u
OUT=$1; shift
R=$(pwd); export TMPDIR=/tmp; D=$(mktemp -d /tmp/pmc.XXXX)
( cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2> $D/err )
F=$(find $D -name '*counter_collection.csv' | head -1)
[ -n "$F" ] && python $R/profiles/pmc_counters.py $F $OUT | grep -E "kernel|k_stats|k_fb|k_prod|k_pair|k_row|k_seg|k_tables" || tail -3 $D/err
