#!/bin/bash
# completion of a pass: the stream's write-value (default) against the completion signal of the last launch itself, polled with hipEventQuery
# (HF_STREAM_STAMP=2: event with timing, 3: without); same box, alternating
HF_STREAM_STAMP=2 python -m pytest tests/test_estep_gpu.py -q -m gpu -x 2>&1 | tail -1
HF_STREAM_STAMP=3 python -m pytest tests/test_estep_gpu.py -q -m gpu -x 2>&1 | tail -1
for i in 1 2 3; do for v in 1 2 3; do
HF_STREAM_STAMP=$v python bench.py --steps 1000 --warmup 1500 --no-cpu-baseline --no-em-run --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('HF_STREAM_STAMP=$v  ms_per_step %.4f' % d['ms_per_step'])"
done; done
for v in 1 2 3; do
HF_STREAM_STAMP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-em-run --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('driver parameters: HF_STREAM_STAMP=$v  ms_per_step %.4f' % d['ms_per_step'])"
done
