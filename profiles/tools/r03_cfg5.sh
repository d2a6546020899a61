timeout 900 python -m pytest tests/test_estep_gpu.py -x -q -k "sparse or wide_coverage or over_dispersed or random_inputs or statistics_modes or multi_region" 2>&1 | tail -5
for c in 5 2 4; do HF_HOST_TRACE=2 python bench.py --config $c --no-cpu-baseline 2> /tmp/e.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('config $c ms_per_step', round(d['ms_per_step'],4), {k: round(v*1000,1) for k,v in d['roofline']['kernel_ms_all'].items()})"; grep "statistics plan" /tmp/e.txt | head -1; done
