for i in 1 2; do for v in r04 "" noguard noinit nc0 all3; do
HF_LIBRARY_VARIANT=$v python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-em-run --event-stride 1 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('lib [%s] k_seg_fb %.1f us' % ('$v' or 'r05', 1e3*d['roofline']['kernel_ms_timed']))"
done; done
