for i in 1 2 3; do for v in "" prio1 prio2 prio4; do
HF_LIBRARY_VARIANT=$v python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-em-run --event-stride 4 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('lib [%s] ms_per_step %.4f k_seg_fb %.1f us' % ('$v' or 'base', d['ms_per_step'], 1e3*d['roofline']['kernel_ms_timed']))"
done; done
