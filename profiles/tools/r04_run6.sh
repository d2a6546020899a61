#!/bin/bash
set -u
mkdir -p gpurun_out
( timeout 1500 python profiles/tools/fuzz_modes.py 20000 1500 2>&1 | tail -8 ) > gpurun_out/r04_fuzz_modes.txt
cat gpurun_out/r04_fuzz_modes.txt
( FUZZ_OPTIONS=1 timeout 1500 python profiles/tools/fuzz_cli.py 12000 300 2>&1 | tail -25 ) > gpurun_out/r04_fuzz_cli.txt
cat gpurun_out/r04_fuzz_cli.txt
bash profiles/tools/cli_wall.sh 2>&1 | head -12
