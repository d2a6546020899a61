#!/bin/bash
# size sweep of the round-4 build (as profiles/tools/r03_scale.sh), the non-headline workloads, and where a workgroup's life goes
# (a -DHF_SEG_TRACE variant built by profiles/tools/build_variants.sh "trace=-DHF_SEG_TRACE"):  bash profiles/tools/r04_scale.sh
set -u
cd "$(dirname "$0")/../.."
one() { python bench.py --no-cpu-baseline --no-em-run --steps 40 "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', 'windows', d['config']['n_windows'], 'ms_per_step', round(d['ms_per_step'],4), 'Gwin/s', round(d['value']/1e9,2), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"; }
for sc in 0.125 0.25 0.5 1 2 4 8; do one --scale $sc; done
for sc in 0.5 0.125; do for ex in ranks chunks; do one --dist-path --no-second-exchange --scale $sc --exchange $ex; done; done
one --config 4
one --config 5
one --config 6
if [ -f flagger_amd/csrc/variants/libhmmflagger_hip.trace.so ]; then
for sc in 1 0.125; do
  echo "== workgroup trace, scale $sc"
  HF_LIBRARY_VARIANT=trace HF_SEG_TRACE_FILE=/tmp/segtrace.bin python bench.py --no-cpu-baseline --no-em-run --steps 30 --warmup 50 --scale $sc > /dev/null 2>&1
  python profiles/tools/seg_trace.py /tmp/segtrace.bin | head -34
done
fi
