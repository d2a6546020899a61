#!/bin/bash
# size sweep of the round-3 build: every contig length of configs[2] x scale (one context), and a rank's share through the
# multi-GPU code path with one RCCL rank (--dist-path), both exchanges:  bash profiles/tools/r03_scale.sh > gpurun_out/r03_scale.txt
set -u
cd "$(dirname "$0")/../.."
one() { python bench.py --no-cpu-baseline --steps 40 "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', 'windows', d['config']['n_windows'], 'ms_per_step', round(d['ms_per_step'],4), 'Gwin/s', round(d['value']/1e9,2), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"; }
for sc in 0.125 0.25 0.5 1 2 4 8; do one --scale $sc; done
for sc in 0.5 0.125; do for ex in ranks chunks; do one --dist-path --no-second-exchange --scale $sc --exchange $ex; done; done
one --config 4
one --config 5
