#!/bin/bash
# round 4, first GPU call: the whole -m gpu suite, the round's evidence set, command-line wall + per-pass times, the residue study
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a_pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04a_pytest.txt
tail -5 gpurun_out/r04a_pytest.txt
timeout 900 bash profiles/tools/r04_collect.sh r04a > gpurun_out/r04a_collect.txt 2>&1
tail -40 gpurun_out/r04a_collect.txt
timeout 300 bash profiles/tools/cli_wall.sh > gpurun_out/r04a_cli_wall.txt 2>&1
cat gpurun_out/r04a_cli_wall.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off profiles/tools/ulp_probe.hip -o /tmp/ulp_probe > /dev/null 2>&1 && /tmp/ulp_probe > gpurun_out/r04_ulp_probe.txt 2>&1
cat gpurun_out/r04_ulp_probe.txt
FUZZ_SHOW=1 timeout 600 python profiles/tools/r04_residue.py > gpurun_out/r04_squarem_residue.txt 2>&1
tail -80 gpurun_out/r04_squarem_residue.txt
