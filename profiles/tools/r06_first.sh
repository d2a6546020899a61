#!/bin/bash
# round 6, first GPU call: (1) hf_create beside a live context, (2) A/B of the segment hand-off (wide accesses x XCD plan), (3) the suite
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
python profiles/tools/r06_create_probe.py > $O/r06_create_probe.txt 2>&1
PROBE_IDLE_S=0.5 PROBE_N=5 python profiles/tools/r06_create_probe.py >> $O/r06_create_probe.txt 2>&1
HF_HOST_THREADS=8 PROBE_N=5 python profiles/tools/r06_create_probe.py >> $O/r06_create_probe.txt 2>&1
tail -30 $O/r06_create_probe.txt
one() {  # name, env...
  local name=$1; shift
  env "$@" python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-em-run --event-stride 4 ${BENCH_EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('[$name] ms_per_step %.4f k_seg_fb %.1f us  all %s' % (d['ms_per_step'], 1e3*d['roofline']['kernel_ms_timed'], {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()}))"
}
{
for i in 1 2 3; do
  one "wide xcd"      HF_SEG_XCD=1
  one "wide plain"    HF_SEG_XCD=0
  one "narrow xcd"    HF_SEG_XCD=1 HF_LIBRARY_VARIANT=narrow
  one "narrow plain"  HF_SEG_XCD=0 HF_LIBRARY_VARIANT=narrow
done
for sc in 0.25 0.125; do for i in 1 2; do
  BENCH_EXTRA="--scale $sc" one "wide xcd scale $sc"     HF_SEG_XCD=1
  BENCH_EXTRA="--scale $sc" one "wide plain scale $sc"   HF_SEG_XCD=0
  BENCH_EXTRA="--scale $sc" one "narrow plain scale $sc" HF_SEG_XCD=0 HF_LIBRARY_VARIANT=narrow
done; done
} > $O/r06_ab_handoff_raw.txt 2>&1
cat $O/r06_ab_handoff_raw.txt
python bench.py --steps 20 --warmup 5 > $O/r06a_bench.json 2> $O/r06a_bench.err; cut -c1-400 $O/r06a_bench.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/r06a_pytest.txt 2>&1; tail -5 $O/r06a_pytest.txt
