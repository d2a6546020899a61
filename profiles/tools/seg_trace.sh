set -u
for X in "$@"; do
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc EXTRA="-DHF_SEG_TRACE $X" > /dev/null 2>&1 || echo build failed
HF_SEG_TRACE_FILE=/tmp/tr.bin python bench.py --no-cpu-baseline --steps 30 2>/tmp/tr.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"
grep "seg trace" /tmp/tr.err; echo "== $X"; python profiles/tools/seg_trace.py /tmp/tr.bin
done
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc > /dev/null 2>&1
