#!/bin/bash
# round 6: per-chunk statistics with four lanes per window (k_stats_win) against one lane per window (k_stats_tile, HF_STATS_TILE=lane)
set -u
cd "$(dirname "$0")/../.."
timeout 900 python profiles/tools/fuzz_modes.py 95000 ${FUZZ_N:-1500} 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
one() { local name=$1; shift
  env "$@" python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-em-run --dist-path --exchange chunks --no-second-exchange ${BENCH_EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('[$name] ms_per_step %.4f LL %.9f all %s' % (d['ms_per_step'], d['loglikelihood_after_last_step'], {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()}))"; }
for i in 1 2; do
  one "four lanes per window (default)" A=1
  one "one lane per window (rounds 1-5)" HF_STATS_TILE=lane
done
for sc in 0.5 0.125; do
  BENCH_EXTRA="--scale $sc" one "scale $sc four lanes per window" A=1
  BENCH_EXTRA="--scale $sc" one "scale $sc one lane per window" HF_STATS_TILE=lane
done
for cfg in 4 5 6; do
  BENCH_EXTRA="--config $cfg" one "config $cfg four lanes per window" A=1
  BENCH_EXTRA="--config $cfg" one "config $cfg one lane per window" HF_STATS_TILE=lane
done
