#!/bin/bash
# round 6: exp restated (hf_exp.h) — new tests, k_tables A/B against the device library's exp, the accelerated fuzz batch with both;
# then the loader at real row density
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "device_exp or retry_pass or xcd_block or static_guard or hand_off" > $O/r06d_pytest_new.txt 2>&1; tail -3 $O/r06d_pytest_new.txt
one() { local name=$1; shift
  env "$@" python bench.py --steps 300 --warmup 150 --no-cpu-baseline --no-em-run --event-stride 4 ${BENCH_EXTRA:-} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('[$name] ms_per_step %.4f  all %s' % (d['ms_per_step'], {a: round(b*1e3,2) for a,b in d['roofline']['kernel_ms_all'].items()}))"; }
{ for i in 1 2 3; do one "hf_exp (glibc algorithm restated)" HF_LIBRARY_VARIANT=glibcexp; one "device library exp" A=1; done
  BENCH_EXTRA="--config 4" one "cfg4 hf_exp" HF_LIBRARY_VARIANT=glibcexp; BENCH_EXTRA="--config 4" one "cfg4 device library exp" A=1
  BENCH_EXTRA="--config 5" one "cfg5 hf_exp" HF_LIBRARY_VARIANT=glibcexp; BENCH_EXTRA="--config 5" one "cfg5 device library exp" A=1
} > $O/r06_ab_exp.txt 2>&1; cat $O/r06_ab_exp.txt
{ echo "# accelerated fuzz runs (every fifth seed of 14000..17999 = 800 runs, FUZZ_OPTIONS=1), product command line against the oracle command line, byte for byte"
  echo "## device library exp (this build)"; FUZZ_OPTIONS=1 FUZZ_ONLY_ACCELERATED=1 python profiles/tools/fuzz_cli.py 14000 4000 2>&1 | tail -40
  echo "## hf_exp: glibc algorithm restated (-DHF_EXP_OCML=0)"; LD_LIBRARY_PATH=$PWD/flagger_amd/csrc/variants/glibcexp:${LD_LIBRARY_PATH:-} FUZZ_OPTIONS=1 FUZZ_ONLY_ACCELERATED=1 python profiles/tools/fuzz_cli.py 14000 4000 2>&1 | tail -100
} > $O/r06_exp_fuzz.txt 2>&1; grep -c DIFFERENT $O/r06_exp_fuzz.txt; grep "^seeds\|^##" $O/r06_exp_fuzz.txt
python profiles/tools/r06_loader_dense.py > $O/r06_loader_dense.txt 2>&1; cat $O/r06_loader_dense.txt
