#!/bin/bash
# round-2 GPU check: the whole -m gpu suite, one bench line, optional A/B of build switches.  bash profiles/tools/r02_check.sh <tag> ["<EXTRA A>" "<EXTRA B>"]
set -u
TAG=${1:-r02}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/$TAG/pytest.log
tail -15 gpurun_out/$TAG/pytest.log
python bench.py --steps 50 --warmup 5 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/$TAG/bench.json")); print("bench", round(d["ms_per_step"],4), {k:round(v*1e3,1) for k,v in d["roofline"]["kernel_ms_all"].items()})
PY
if [ $# -ge 3 ]; then bash profiles/ab.sh "$2" "$3" 2>&1 | tee gpurun_out/$TAG/ab.txt; fi
