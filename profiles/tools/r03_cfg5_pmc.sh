export BENCH_ARGS="--config 5" HF_RS_BPW=${HF_RS_BPW:-4}
bash profiles/pmc_pass.sh gpurun_out/cfg5_sq_a.json SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
bash profiles/pmc_pass.sh gpurun_out/cfg5_sq_b.json SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
R=$(pwd); export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/kt5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -o kt -- python $R/bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 )
F=$(find /tmp/kt5 -name '*kernel_stats.csv' | head -1); cut -c1-110 $F | head -12
