#!/bin/bash
# round 6: the split hf_create — identical plans (1000 shapes, digest per seed against the library before the split), timing, full suite
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
python profiles/tools/r06_plan_identity.py 60000 1000 > $O/r06_plan_new.txt 2> $O/r06_plan_new.err &
HF_LIBRARY_VARIANT=precreate python profiles/tools/r06_plan_identity.py 60000 1000 > $O/r06_plan_old.txt 2> $O/r06_plan_old.err
wait
wc -l $O/r06_plan_new.txt $O/r06_plan_old.txt; tail -2 $O/r06_plan_new.err
if diff -q $O/r06_plan_new.txt $O/r06_plan_old.txt > /dev/null; then echo "PLANS IDENTICAL: $(wc -l < $O/r06_plan_new.txt) shapes, every digest equal"; else echo "PLANS DIFFER"; diff $O/r06_plan_new.txt $O/r06_plan_old.txt | head -10; fi
HF_PARTS_TRACE=1 PROBE_N=3 python profiles/tools/r06_create_probe.py > $O/r06_parts_trace2.txt 2>&1
python profiles/tools/r06_create_probe.py > $O/r06_create_probe_b.txt 2>&1
grep "context\|median" $O/r06_create_probe_b.txt | cut -c1-330
python bench.py --steps 20 --warmup 5 > $O/r06c_bench.json 2> $O/r06c_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06c_bench.json') if l.startswith('{"metric"')][-1])
print("ms_per_step", d["ms_per_step"], "k_seg_fb", d["roofline"]["kernel_ms_timed"])
e=d["em_run"]; print({k: e[k] for k in e if k.startswith("hf_create")})
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r06c_pytest.txt 2>&1; grep -n "passed\|failed\|skipped" $O/r06c_pytest.txt | tail -3
