#!/bin/bash
# number of sub-passes again, after the scale store left the EM pass: scale 2 (3.05 M windows) and 4 (6.1 M) with 1, auto and values around it
for sc in 1.5 2 4; do for S in auto 1 2 3 4; do
if [ $S = auto ]; then unset HF_SUBPASSES; else export HF_SUBPASSES=$S; fi
python bench.py --scale $sc --steps 200 --warmup 60 --no-cpu-baseline --no-em-run --event-stride 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc HF_SUBPASSES=$S sub_passes %s windows %d ms_per_step %.4f = %.1f ps per window' % (d['roofline'].get('sub_passes'), d['config']['n_windows'], d['ms_per_step'], d['ms_per_step']*1e9/d['config']['n_windows']), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"
done; done
