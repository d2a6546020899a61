#!/bin/bash
# round 6: two micro-variants of k_seg_fb, same box, alternating: staged records 80 bytes apart (no LDS bank conflicts) / v_rcp + two Newton steps
set -u
cd "$(dirname "$0")/../.."
one() { local name=$1; shift
  env "$@" python bench.py --steps 300 --warmup 150 --no-cpu-baseline --no-em-run --event-stride 4 ${BENCH_EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('[$name] ms_per_step %.4f k_seg_fb %.2f us  all %s' % (d['ms_per_step'], 1e3*d['roofline']['kernel_ms_timed'], {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()}))"; }
timeout 600 python -m pytest tests/test_estep_gpu.py -m gpu -x -q -k "small_diploid or cfg1 or multi_region or ragged or full_size_cfg2" 2>&1 | tail -2
HF_LIBRARY_VARIANT=fastrcp timeout 600 python -m pytest tests/test_estep_gpu.py -m gpu -x -q -k "small_diploid or cfg1 or multi_region or ragged or full_size_cfg2" 2>&1 | tail -2
for i in 1 2 3; do
  one "records padded (default)" A=1
  one "records 64 B apart (rounds 3-5)" HF_LIBRARY_VARIANT=nopad
  one "padded + fast reciprocal" HF_LIBRARY_VARIANT=fastrcp
  one "64 B apart + fast reciprocal" HF_LIBRARY_VARIANT=nopad_fastrcp
done
for sc in 0.25 0.125; do for i in 1 2; do
  BENCH_EXTRA="--scale $sc" one "scale $sc padded" A=1
  BENCH_EXTRA="--scale $sc" one "scale $sc 64 B apart" HF_LIBRARY_VARIANT=nopad
  BENCH_EXTRA="--scale $sc" one "scale $sc padded + fast reciprocal" HF_LIBRARY_VARIANT=fastrcp
done; done
