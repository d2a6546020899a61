# cached row blocks (hf_seg.h, round 5): auto-chosen nc against nc = 0, per-kernel times from HIP events around every kernel
for sc in 0.125 0.25 0.5 1; do for nc in 0 auto; do
if [ $nc = auto ]; then unset HF_SEG_CACHED_STEPS; else export HF_SEG_CACHED_STEPS=$nc; fi
python bench.py --scale $sc --steps 300 --warmup 100 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc nc=$nc ms_per_step %.4f' % d['ms_per_step'])"
python bench.py --scale $sc --steps 40 --warmup 20 --no-cpu-baseline --event-stride 1 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('    k_seg_fb us %.1f (median of %d)' % (1e3*d['roofline']['kernel_ms_timed'], d['roofline']['n_samples']), {k: round(1e3*v,1) for k,v in d['roofline']['kernel_ms_all'].items()})"
done; done
