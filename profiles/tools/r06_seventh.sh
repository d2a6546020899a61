#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
bash profiles/tools/r06_final.sh > gpurun_out/r06h_final.txt 2>&1
tail -60 gpurun_out/r06h_final.txt | cut -c1-400
timeout 1800 python -m pytest tests -m gpu -q -rs > gpurun_out/r06h_pytest.txt 2>&1; tail -12 gpurun_out/r06h_pytest.txt | cut -c1-300
