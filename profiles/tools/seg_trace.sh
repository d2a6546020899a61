#!/bin/bash
# timing build of k_seg_fb with s_memtime stamps (-DHF_SEG_TRACE, hf_seg.h; results unchanged, the stamps cost a little):
#   [BENCH_ARGS="--scale 0.125"] bash profiles/tools/seg_trace.sh ["<extra build switches>" ...]
set -u
cd "$(dirname "$0")/../.."
for X in "${@:-}"; do
  touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc EXTRA="-DHF_SEG_TRACE $X" > /dev/null 2>&1 || echo build failed
  HF_SEG_TRACE_FILE=/tmp/segtrace.bin python bench.py --no-cpu-baseline --steps 30 ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$X]', round(d['ms_per_step'],4), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"
  python profiles/tools/seg_trace.py /tmp/segtrace.bin
done
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc > /dev/null 2>&1
