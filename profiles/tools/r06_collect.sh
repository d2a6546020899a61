#!/bin/bash
# Evidence of a round-6 snapshot, on a GPU box from the repo root:  bash profiles/tools/r06_collect.sh <tag>
#   <tag>_bench.json           the bench line with the DRIVER's parameters (--steps 20 --warmup 5): roofline (median of >= 8 event samples),
#                              roofline_fp64, em_run (warm process + cold command line), cpu_baseline
#   <tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command (per-kernel calls / average us); <tag>_timeline.txt
#   <tag>_pmc_traffic.json     FETCH_SIZE / WRITE_SIZE, two separate passes, tagged (_meta.tag) so that bench.py finds the passes below
#   <tag>_pmc_sq_a/b.json      SQ instruction / wait / busy counters
#   <tag>_pmc_fp64.json        SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64: the flops of roofline_fp64
set -u
TAG=${1:-r06a}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
QUIET="--no-cpu-baseline --no-em-run"
( cd /tmp && rm -rf /tmp/prof_kt && rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $R/bench.py --steps 20 --warmup 5 $QUIET > $OUT/${TAG}_rocprof_bench.json 2> $OUT/${TAG}_rocprof.err )
DB=$(find /tmp/prof_kt -name '*_results.db' | head -1)
python profiles/summarize.py $DB $OUT/${TAG}_kernel_stats.csv > /dev/null
( cd /tmp && rm -rf /tmp/kt2 && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -o kt -- python $R/bench.py --steps 20 --warmup 5 $QUIET > /dev/null 2>&1 )
F=$(find /tmp/kt2 -name '*kernel_trace.csv' | head -1)
python profiles/tools/timeline.py $F $OUT/${TAG}_timeline.txt > /dev/null
PMC="python $R/bench.py --steps 3 --warmup 1 $QUIET"
( cd /tmp && rm -rf /tmp/prof_rd /tmp/prof_wr
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_rd -o rd -- $PMC > /dev/null 2> $OUT/${TAG}_pmc_rd.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_wr -o wr -- $PMC > /dev/null 2> $OUT/${TAG}_pmc_wr.err )
RD=$(find /tmp/prof_rd -name '*counter_collection.csv' | head -1)
WR=$(find /tmp/prof_wr -name '*counter_collection.csv' | head -1)
python profiles/pmc_summary.py $RD $WR $OUT/${TAG}_pmc_traffic.json $TAG > /dev/null
export BENCH_ARGS="--no-em-run"
bash profiles/pmc_pass.sh $OUT/${TAG}_pmc_sq_a.json SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR > /dev/null 2>&1
bash profiles/pmc_pass.sh $OUT/${TAG}_pmc_sq_b.json SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE > /dev/null 2>&1
bash profiles/pmc_pass.sh $OUT/${TAG}_pmc_fp64.json SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 > /dev/null 2>&1
cut -c1-1500 $OUT/${TAG}_bench.json; echo; cut -c1-90 $OUT/${TAG}_kernel_stats.csv; cat $OUT/${TAG}_timeline.txt | tail -12
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_pmc_traffic.json"))
for k,v in d.items():
    if not k.startswith("_"): print(k[:30], round(v["read_bytes_per_launch"]/1e6,1), round(v["write_bytes_per_launch"]/1e6,1))
try:
    for k,v in json.load(open("$OUT/${TAG}_pmc_fp64.json")).items(): print(k[:30], {a[14:]: round(b) for a,b in v.items()})
except Exception as e: print("fp64 pass:", e)
PY
# round 6: the write requests of k_seg_fb (are the 64-byte record stores merged into whole lines on their way out of the L2?)
bash profiles/pmc_pass.sh $OUT/${TAG}_pmc_wrreq.json TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum > /dev/null 2>&1
bash profiles/pmc_pass.sh $OUT/${TAG}_pmc_tcc.json TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum TCC_WRITEBACK_sum > /dev/null 2>&1
