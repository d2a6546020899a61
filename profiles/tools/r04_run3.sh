#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r04d_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04d_pytest.txt
tail -8 gpurun_out/r04d_pytest.txt
for i in 1 2; do
python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-em-run --event-stride 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stride 1:', d['ms_per_step'], d['roofline']['kernel_ms_samples'])"
python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-em-run --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no events:', d['ms_per_step'])"
done
timeout 900 bash profiles/tools/r04_collect.sh r04d > gpurun_out/r04d_collect.txt 2>&1
tail -30 gpurun_out/r04d_collect.txt | cut -c1-1800
