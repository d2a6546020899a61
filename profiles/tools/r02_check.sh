#!/bin/bash
# round-2 GPU check: the whole -m gpu suite, the bench line, the worst-case table workload.  bash profiles/tools/r02_check.sh <tag>
set -u
TAG=${1:-r02}
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/$TAG/pytest.log
tail -15 gpurun_out/$TAG/pytest.log
python bench.py --steps 50 --warmup 5 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
python bench.py --steps 30 --warmup 3 --config 5 --no-cpu-baseline > gpurun_out/$TAG/bench_cfg5.json 2>> gpurun_out/$TAG/bench.err
python bench.py --steps 30 --warmup 3 --config 4 --no-cpu-baseline > gpurun_out/$TAG/bench_cfg4.json 2>> gpurun_out/$TAG/bench.err
python - <<PY
import json
for f in ("bench", "bench_cfg4", "bench_cfg5"):
    try:
        d=json.load(open("gpurun_out/$TAG/%s.json" % f)); print(f, round(d["ms_per_step"],4), d["config"]["statistics"], {k:round(v*1e3,1) for k,v in d["roofline"]["kernel_ms_all"].items()})
    except Exception as e: print(f, "failed", e)
PY
cat gpurun_out/neartie.json 2>/dev/null
