#!/bin/bash
# size sweep on the final build: ms per EM step (1 000 steps behind 1 500 up to x 1: behind the runtime's one-time stall; sub-passes beyond); two rounds
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for sc in 0.125 0.25 0.5 1; do python bench.py --scale $sc --steps 1000 --warmup 1500 --no-cpu-baseline --no-em-run --event-stride 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc ms_per_step %.4f  k_seg_fb %.1f us  cached blocks %s sub-passes %s' % (d['ms_per_step'], 1e3*d['roofline']['kernel_ms_timed'], d['roofline'].get('cached_row_blocks'), d['roofline'].get('sub_passes')), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"; done
for sc in 2 4 8; do python bench.py --scale $sc --steps 200 --warmup 60 --no-cpu-baseline --no-em-run --event-stride 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc ms_per_step %.4f  k_seg_fb %.1f us  cached blocks %s sub-passes %s' % (d['ms_per_step'], 1e3*d['roofline']['kernel_ms_timed'], d['roofline'].get('cached_row_blocks'), d['roofline'].get('sub_passes')), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"; done
done
