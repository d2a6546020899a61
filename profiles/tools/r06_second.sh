#!/bin/bash
# round 6, second GPU call: (1) the driver's bench command with hf_create x 9 + phases, (2) where the XCD plan loses (workgroup trace),
# (3) write-request counters of k_seg_fb (are the 64-byte record stores merged into whole lines?)
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/r06b_bench.json 2> $O/r06b_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06b_bench.json') if l.startswith('{"metric"')][-1])
print("ms_per_step", d["ms_per_step"], "k_seg_fb", d["roofline"]["kernel_ms_timed"])
e=d["em_run"]; print({k: e[k] for k in e if k.startswith("hf_create")})
PY
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r06b_bench2.json 2>/dev/null
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06b_bench2.json') if l.startswith('{"metric"')][-1])
e=d["em_run"]; print("second process:", {k: e[k] for k in e if k.startswith("hf_create_ms")})
PY
{
for x in 0 1; do
  echo "# HF_SEG_XCD=$x"
  HF_SEG_XCD=$x HF_LIBRARY_VARIANT=trace HF_SEG_TRACE_FILE=/tmp/segtrace.bin python bench.py --no-cpu-baseline --no-em-run --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"
  python profiles/tools/seg_trace.py /tmp/segtrace.bin
done
} > $O/r06_seg_trace_xcd.txt 2>&1
head -40 $O/r06_seg_trace_xcd.txt
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_WR[A-Z0-9_]*\|TCC_EA0_RD[A-Z0-9_]*" | sort -u | tr '\n' ' ' ) > $O/r06_tcc_counters_avail.txt; cat $O/r06_tcc_counters_avail.txt; echo
export BENCH_ARGS="--no-em-run"
bash profiles/pmc_pass.sh $O/r06_pmc_wrreq.json TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum 2>&1 | tail -8
bash profiles/pmc_pass.sh $O/r06_pmc_wrreq2.json TCC_EA0_WRREQ_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_STALL_sum 2>&1 | tail -8
bash profiles/pmc_pass.sh $O/r06_pmc_tcc.json TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum TCC_WRITEBACK_sum 2>&1 | tail -8
