for sc in 0.125 0.25 0.5 1 2; do for v in 0 1; do
HF_TAB_FUSED=$v python bench.py --scale $sc --steps 300 --warmup 100 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc TAB_FUSED=$v ms_per_step %.4f' % d['ms_per_step'])"
done; done
