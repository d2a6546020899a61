#!/bin/bash
# EXPERIMENT: two sub-passes on two streams (HF_SUBPASSES=2 HF_OVERLAP=1|2|3: second stream default / low / high priority) against one sub-pass
for i in 1 2 3; do for v in "1 0" "2 0" "2 1" "2 2" "2 3"; do set -- $v
HF_SUBPASSES=$1 HF_OVERLAP=$2 python bench.py --steps 1000 --warmup 1500 --no-cpu-baseline --no-em-run --event-stride 4 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('subpasses $1 overlap $2  ms_per_step %.4f' % d['ms_per_step'], {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"
done; done
