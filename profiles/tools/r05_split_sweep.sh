#!/bin/bash
# segment length sweep (HF_SEG_SPLIT windows per segment at most) at small sizes, with and without cached row blocks
for sc in 0.125 0.25 0.5; do for sp in 512 384 256 192 128; do for nc in 0 L; do
  L=$(( (sp + 63) / 64 )); if [ $nc = L ]; then ncv=$L; else ncv=0; fi
  HF_SEG_SPLIT=$sp HF_SEG_CACHED_STEPS=$ncv python bench.py --scale $sc --steps 600 --warmup 1500 --no-cpu-baseline --no-em-run --event-stride 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc split $sp nc $ncv  ms_per_step %.4f' % d['ms_per_step'], {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"
done; done; done
