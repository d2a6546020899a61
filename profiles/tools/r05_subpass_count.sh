for S in 1 3 4 5 6 8; do
HF_SUBPASSES=$S python bench.py --scale 4 --steps 150 --warmup 100 --no-cpu-baseline --no-kernel-events --no-em-run 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale 4 HF_SUBPASSES=$S ms_per_step %.4f  %.1f ps/window' % (d['ms_per_step'], d['ms_per_step']*1e9/d['config']['n_windows']))"
done
for S in 1 2 3; do
HF_SUBPASSES=$S python bench.py --scale 2 --steps 200 --warmup 100 --no-cpu-baseline --no-kernel-events --no-em-run 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale 2 HF_SUBPASSES=$S ms_per_step %.4f  %.1f ps/window' % (d['ms_per_step'], d['ms_per_step']*1e9/d['config']['n_windows']))"
done
HF_SUBPASSES=4 python bench.py --scale 4 --steps 40 --warmup 40 --no-cpu-baseline --event-stride 1 --no-em-run 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_all'])"
HF_SUBPASSES=1 python bench.py --scale 4 --steps 40 --warmup 40 --no-cpu-baseline --event-stride 1 --no-em-run 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_all'])"
