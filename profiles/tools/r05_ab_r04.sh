# same-box A/B of the round-4 library (git 7a26c49, built as csrc/variants/libhmmflagger_hip.r04.so) against this round's, the driver's command
for i in 1 2 3; do for v in r04 ""; do
HF_LIBRARY_VARIANT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-em-run 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('lib [%s] ms_per_step %.4f  k_seg_fb %.1f us  others' % ('$v' or 'r05', d['ms_per_step'], 1e3*d['roofline']['kernel_ms_timed']), {k: round(1e3*v,1) for k,v in d['roofline']['kernel_ms_all'].items() if k != 'k_seg_fb'})"
done; done
for v in r04 ""; do HF_LIBRARY_VARIANT=$v python bench.py --steps 1000 --warmup 1000 --no-cpu-baseline --no-em-run --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('lib [%s] 1000 steps behind 1000, no events: ms_per_step %.4f' % ('$v' or 'r05', d['ms_per_step']))"; done
