timeout 600 python -m pytest tests/test_neartie_gpu.py -m gpu -x -q 2>&1 | tail -8; cat gpurun_out/neartie.json
for X in "" "-DHF_TABLE_JOBS_PER_BLOCK=8" "-DHF_TABLE_JOBS_PER_BLOCK=16" "-DHF_TABLE_JOBS_PER_BLOCK=4"; do
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc EXTRA="$X" > /dev/null 2>&1 || echo build failed
for cfg in 2 5; do python bench.py --no-cpu-baseline --steps 30 --config $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$X] cfg$cfg', round(d['ms_per_step'],4), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"; done
done
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc > /dev/null 2>&1
