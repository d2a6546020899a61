#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_estep_gpu.py tests/test_multi_gpu.py -m gpu -x -q > gpurun_out/r04e_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04e_pytest.txt
tail -4 gpurun_out/r04e_pytest.txt
ROUNDS=3 timeout 900 bash profiles/tools/r04_ab_variants.sh noRECSWZ new3 > gpurun_out/r04e_ab_variants.txt 2>&1
grep -v "simple_timer\|^$" gpurun_out/r04e_ab_variants.txt
bash profiles/tools/cli_wall.sh 2>&1 | head -12
