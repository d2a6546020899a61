#!/bin/bash
# k_seg_fb without the per-window scale store in EM passes (the getters run the kernel again with it) against the library that always stores it (variant `scales`)
python -m pytest tests/test_estep_gpu.py tests/test_multi_gpu.py tests/test_shim_gpu.py -q -m gpu -x 2>&1 | tail -1
for i in 1 2 3; do for v in "" scales; do
HF_LIBRARY_VARIANT=$v python bench.py --steps 1000 --warmup 1500 --no-cpu-baseline --no-em-run --event-stride 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('lib [%s]  ms_per_step %.4f  k_seg_fb %.1f us' % ('$v' or 'no scale store', d['ms_per_step'], 1e3*d['roofline']['kernel_ms_timed']))"
done; done
for i in 1 2; do for v in "" scales; do
HF_LIBRARY_VARIANT=$v python bench.py --scale 4 --steps 300 --warmup 100 --no-cpu-baseline --no-em-run --event-stride 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale 4: lib [%s]  ms_per_step %.4f  k_seg_fb %.1f us' % ('$v' or 'no scale store', d['ms_per_step'], 1e3*d['roofline']['kernel_ms_timed']))"
done; done
