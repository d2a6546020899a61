#!/bin/bash
# does the driver's short run (20 steps behind 5) pay for a GPU clock that is still ramping?  hf_create with HF_PREWARM_US=n busy microseconds ahead
set -u
cd "$(dirname "$0")/../.."
for rep in 1 2 3; do for us in 0 1000 3000 10000 50000; do
  HF_PREWARM_US=$us python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-em-run 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); r=d['roofline']; print('prewarm $us us: ms_per_step %.4f  k_seg_fb median %.1f us (min %.1f max %.1f)' % (d['ms_per_step'], 1e3*r['kernel_ms_timed'], 1e3*r['kernel_ms_samples']['min'], 1e3*r['kernel_ms_samples']['max']))"
done; done
rocm-smi --showclocks 2>/dev/null | head -20
