#!/bin/bash
# round 6: where the step's time outside its kernels goes — start stamps of the four launches, no profiler (a -DHF_KSTAMP build of the library)
set -u
cd "$(dirname "$0")/../.."
rm -f /tmp/kstamp.bin
HF_LIBRARY_VARIANT=kstamp HF_KSTAMP_FILE=/tmp/kstamp.bin python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-em-run --creates 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('ms_per_step %.4f  without events %s  kernels %s' % (d['ms_per_step'], d.get('ms_per_step_without_kernel_events'), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()}))"
python profiles/tools/r06_kstamp.py /tmp/kstamp.bin
rm -f /tmp/kstamp.bin
echo "--- settled: 2000 steps behind 1000, no kernel events ---"
HF_LIBRARY_VARIANT=kstamp HF_KSTAMP_FILE=/tmp/kstamp.bin python bench.py --steps 2000 --warmup 1000 --no-cpu-baseline --no-em-run --creates 0 --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('ms_per_step %.4f' % d['ms_per_step'])"
python profiles/tools/r06_kstamp.py /tmp/kstamp.bin -60 -14 | tail -7
