#!/bin/bash
# soak of the one-launch hand-off on the round-4 kernel: many consecutive EM steps, no wait may be given up (a "falls back" line on stderr)
set -u
cd "$(dirname "$0")/../.."
for spec in "2 100000" "4 20000" "5 20000" "6 20000"; do
  set -- $spec
  python bench.py --config $1 --steps $2 --warmup 10 --no-cpu-baseline --no-em-run --no-kernel-events 2> /tmp/soak.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $1: $2 steps, ms_per_step', round(d['ms_per_step'],4))"
  echo "   hand-off time-outs: $(grep -c 'falls back' /tmp/soak.err)"
done
