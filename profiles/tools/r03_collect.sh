#!/bin/bash
# evidence of a round-3 snapshot: bash profiles/tools/r02_collect.sh <tag>  (bench, kernel trace + stats + timeline, FETCH / WRITE passes, SQ counters)
set -u
TAG=${1:-r03i}
bash profiles/collect.sh $TAG > /dev/null 2>&1
R=$(pwd); export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o kt -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 )
F=$(find /tmp/kt2 -name '*kernel_trace.csv' | head -1)
python profiles/tools/timeline.py $F gpurun_out/${TAG}_timeline.txt > /dev/null
bash profiles/pmc_pass.sh gpurun_out/${TAG}_pmc_sq_a.json SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR > /dev/null 2>&1
bash profiles/pmc_pass.sh gpurun_out/${TAG}_pmc_sq_b.json SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > /dev/null 2>&1
cat gpurun_out/${TAG}_bench.json | cut -c1-300; cat gpurun_out/${TAG}_kernel_stats.csv | cut -c1-80; cat gpurun_out/${TAG}_timeline.txt; python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_pmc_traffic.json"))
for k,v in d.items(): print(k[:30], round(v["read_bytes_per_launch"]/1e6,1), round(v["write_bytes_per_launch"]/1e6,1))
PY
