for sc in 1 2 4 8; do
python bench.py --scale $sc --steps 40 --warmup 60 --no-cpu-baseline --event-stride 1 --no-em-run 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc auto: ms_per_step %.4f' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['kernel_ms_all'].items()})"
done
