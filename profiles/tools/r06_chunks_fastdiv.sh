#!/bin/bash
# round 6: k_stats_tile (per-chunk statistics: --exchange chunks, the default of hmm_flagger --gpus N) with prepared reciprocals instead of IEEE divisions
set -u
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests -m gpu -x -q -k "chunk or exchange or loopback or sharded or multi" 2>&1 | tail -2
one() { local name=$1; shift
  env "$@" python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-em-run --dist-path --exchange chunks --no-second-exchange ${BENCH_EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('[$name] ms_per_step %.4f all %s' % (d['ms_per_step'], {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()}))"; }
for i in 1 2; do
  one "prepared reciprocals (default)" A=1
  one "IEEE divisions (rounds 1-5)" HF_LIBRARY_VARIANT=olddiv
done
for sc in 0.5 0.125; do
  BENCH_EXTRA="--scale $sc" one "scale $sc prepared reciprocals" A=1
  BENCH_EXTRA="--scale $sc" one "scale $sc IEEE divisions" HF_LIBRARY_VARIANT=olddiv
done
