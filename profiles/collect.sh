#!/bin/bash
# Collect the round's evidence on the GPU box: bench line, rocprofv3 kernel trace + stats, and the two PMC passes
# (FETCH_SIZE and WRITE_SIZE need separate passes: MI355X_MICROARCH.md, TCC counter budget).
# Usage (from the repo root, under gpurun): bash profiles/collect.sh <tag>     -> gpurun_out/<tag>_*
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 50 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp
BENCH="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rm -rf /tmp/prof_kt /tmp/prof_rd /tmp/prof_wr
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $BENCH > $OUT/${TAG}_rocprof_bench.json 2> $OUT/${TAG}_rocprof.err
DB=$(find /tmp/prof_kt -name '*_results.db' | head -1)
python $REPO/profiles/summarize.py $DB $OUT/${TAG}_kernel_stats.csv
PMC="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_rd -o rd -- $PMC > /dev/null 2> $OUT/${TAG}_pmc_rd.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_wr -o wr -- $PMC > /dev/null 2> $OUT/${TAG}_pmc_wr.err
RD=$(find /tmp/prof_rd -name '*counter_collection.csv' | head -1)
WR=$(find /tmp/prof_wr -name '*counter_collection.csv' | head -1)
python $REPO/profiles/pmc_summary.py $RD $WR $OUT/${TAG}_pmc_traffic.json
cat $OUT/${TAG}_bench.json
cat $OUT/${TAG}_kernel_stats.csv
