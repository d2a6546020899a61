#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04c_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04c_pytest.txt
tail -5 gpurun_out/r04c_pytest.txt
timeout 1500 bash profiles/tools/r04_ab_variants.sh new1 new2 noIDBCAST noTRASH withVECSUF > gpurun_out/r04c_ab_variants.txt 2>&1
grep -v "simple_timer\|^$" gpurun_out/r04c_ab_variants.txt
timeout 300 bash profiles/tools/cli_wall.sh > gpurun_out/r04c_cli_wall.txt 2>&1
cat gpurun_out/r04c_cli_wall.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04c_bench.json 2> gpurun_out/r04c_bench.err; cut -c1-2500 gpurun_out/r04c_bench.json
python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-em-run --event-stride 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stride 1:', d['ms_per_step'], d['roofline']['kernel_ms_samples'])"
python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-em-run --no-kernel-events 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no events:', d['ms_per_step'])"
