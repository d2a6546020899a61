#!/bin/bash
# rocprofv3 kernel statistics of the non-headline workloads (round 6): bash profiles/tools/r04_cfg_stats.sh  -> gpurun_out/r06i_cfg<N>_kernel_stats.csv
set -u
R=$(pwd); export TMPDIR=/tmp
for c in 4 5 6; do
  ( cd /tmp && rm -rf /tmp/ktc && rocprofv3 --kernel-trace --stats -d /tmp/ktc -o kt -- python $R/bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-em-run > $R/gpurun_out/r06i_cfg${c}_bench.json 2> /dev/null )
  DB=$(find /tmp/ktc -name '*_results.db' | head -1)
  python $R/profiles/summarize.py $DB $R/gpurun_out/r06i_cfg${c}_kernel_stats.csv > /dev/null
  echo "== config $c"; head -6 $R/gpurun_out/r06i_cfg${c}_kernel_stats.csv | cut -c1-40,180-260
done
