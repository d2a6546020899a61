#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
one() { local name=$1; shift
  env "$@" python bench.py --steps 300 --warmup 150 --no-cpu-baseline --no-em-run --event-stride 4 ${BENCH_EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('[$name] ms_per_step %.4f  all %s' % (d['ms_per_step'], {a: round(b*1e3,2) for a,b in d['roofline']['kernel_ms_all'].items()}))"; }
{ for i in 1 2 3; do one "hf_exp (glibc algorithm restated)" HF_LIBRARY_VARIANT=glibcexp; one "device library exp" A=1; done
  BENCH_EXTRA="--config 4" one "cfg4 hf_exp" HF_LIBRARY_VARIANT=glibcexp; BENCH_EXTRA="--config 4" one "cfg4 device library exp" A=1
  BENCH_EXTRA="--config 5" one "cfg5 hf_exp" HF_LIBRARY_VARIANT=glibcexp; BENCH_EXTRA="--config 5" one "cfg5 device library exp" A=1
} > $O/r06_ab_exp.txt 2>&1; cat $O/r06_ab_exp.txt
python profiles/tools/r06_loader_dense.py > $O/r06_loader_dense.txt 2>&1; cat $O/r06_loader_dense.txt
timeout 1800 python -m pytest tests -m gpu -q > $O/r06e_pytest.txt 2>&1; grep -n "passed\|failed\|skipped" $O/r06e_pytest.txt | tail -3; grep -n "^FAILED" $O/r06e_pytest.txt | head
