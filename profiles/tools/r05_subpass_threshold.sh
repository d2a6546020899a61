for sc in 1.6 1.7 1.85; do for S in 1 2; do
HF_SUBPASSES=$S python bench.py --scale $sc --steps 200 --warmup 60 --no-cpu-baseline --no-em-run --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc HF_SUBPASSES=$S windows %d ms_per_step %.4f = %.1f ps per window' % (d['config']['n_windows'], d['ms_per_step'], d['ms_per_step']*1e9/d['config']['n_windows']))"
done; done
