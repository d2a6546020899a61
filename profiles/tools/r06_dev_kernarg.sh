#!/bin/bash
# round 6: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1: the dispatch reads them locally instead of over PCIe) against the default
set -u
cd "$(dirname "$0")/../.."
one() { local name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-em-run --creates 0 ${BENCH_EXTRA} 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('[$name] ms_per_step %.4f without events %s' % (d['ms_per_step'], d.get('ms_per_step_without_kernel_events')))"; }
for i in 1 2 3; do
  BENCH_EXTRA="--steps 300 --warmup 150 --no-kernel-events" one "settled, kernarg default" HIP_FORCE_DEV_KERNARG=0
  BENCH_EXTRA="--steps 300 --warmup 150 --no-kernel-events" one "settled, kernarg in device memory" HIP_FORCE_DEV_KERNARG=1
done
for i in 1 2 3; do
  BENCH_EXTRA="--steps 20 --warmup 5" one "driver regime, kernarg default" HIP_FORCE_DEV_KERNARG=0
  BENCH_EXTRA="--steps 20 --warmup 5" one "driver regime, kernarg in device memory" HIP_FORCE_DEV_KERNARG=1
done
