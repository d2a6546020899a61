#!/bin/bash
# round 6: k_seg_fb with the grid's parts started HF_SEG_STAGGER cycles apart (the phases of the resident wavefronts out of step)
set -u
cd "$(dirname "$0")/../.."
one() { local name=$1; shift
  env "$@" python bench.py --steps 300 --warmup 150 --no-cpu-baseline --no-em-run --event-stride 4 ${BENCH_EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('[$name] ms_per_step %.4f k_seg_fb %.2f us  all %s' % (d['ms_per_step'], 1e3*d['roofline']['kernel_ms_timed'], {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()}))"; }
for i in 1 2; do
  one "all together (default)" A=1
  for v in ${VARIANTS:-st5k st10k st15k st20k st3x5k st3x8k st4x5k}; do one "$v" HF_LIBRARY_VARIANT=$v; done
done
