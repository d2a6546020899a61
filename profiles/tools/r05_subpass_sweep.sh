# sub-passes (hf_ctx::SubPass, round 5): inputs past the Infinity Cache with the automatic number of sub-passes against one pass over everything
for sc in 1 2 4 8; do for S in 1 auto; do
if [ $S = auto ]; then unset HF_SUBPASSES; else export HF_SUBPASSES=$S; fi
python bench.py --scale $sc --steps 100 --warmup 30 --no-cpu-baseline --no-kernel-events --no-em-run 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc HF_SUBPASSES=$S windows %d ms_per_step %.4f  = %.1f ps per window and pass, %.2f G windows/s' % (d['config']['n_windows'], d['ms_per_step'], d['ms_per_step']*1e9/d['config']['n_windows'], d['value']/1e9))"
done; done
