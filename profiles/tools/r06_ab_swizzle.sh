#!/bin/bash
# round 6: staged records of k_seg_fb — 80 bytes apart (default) against 64 bytes apart with the pieces XOR-swizzled (conflict-free on both sides)
set -u
cd "$(dirname "$0")/../.."
one() { local name=$1; shift
  env "$@" python bench.py --steps 300 --warmup 150 --no-cpu-baseline --no-em-run --event-stride 4 ${BENCH_EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('[$name] ms_per_step %.4f k_seg_fb %.2f us  all %s' % (d['ms_per_step'], 1e3*d['roofline']['kernel_ms_timed'], {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()}))"; }
HF_LIBRARY_VARIANT=swz timeout 600 python -m pytest tests/test_estep_gpu.py -m gpu -x -q -k "small_diploid or cfg1 or multi_region or ragged or full_size_cfg2" 2>&1 | tail -2
for i in 1 2 3 4; do
  one "80 B apart (default)" A=1
  one "64 B apart, swizzled" HF_LIBRARY_VARIANT=swz
  one "64 B apart (rounds 3-5)" HF_LIBRARY_VARIANT=nopad
done
for i in 1 2; do
  BENCH_EXTRA="--scale 0.25" one "scale 0.25 80 B" A=1
  BENCH_EXTRA="--scale 0.25" one "scale 0.25 swizzled" HF_LIBRARY_VARIANT=swz
done
