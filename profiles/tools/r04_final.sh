#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r04g_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04g_pytest.txt
tail -3 gpurun_out/r04g_pytest.txt
timeout 900 bash profiles/tools/r04_collect.sh r04g > gpurun_out/r04g_collect.txt 2>&1
tail -25 gpurun_out/r04g_collect.txt | cut -c1-400
cp gpurun_out/r04g_pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/r04g_pmc_sq_b.json gpurun_out/r04g_pmc_fp64.json profiles/
python bench.py --steps 20 --warmup 5 > gpurun_out/r04g_bench.json 2> gpurun_out/r04g_bench.err
cut -c1-300 gpurun_out/r04g_bench.json
python bench.py > gpurun_out/r04g_bench_default.json 2>/dev/null; cut -c1-300 gpurun_out/r04g_bench_default.json
bash profiles/tools/cli_wall.sh > gpurun_out/r04g_cli_wall.txt 2>&1; head -12 gpurun_out/r04g_cli_wall.txt
python profiles/tools/nb_step_time.py 2>&1 | tail -3; grep -c passed gpurun_out/r04g_pytest.txt; grep -E "passed|failed" gpurun_out/r04g_pytest.txt | tail -2
