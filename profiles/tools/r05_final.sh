#!/bin/bash
# round-5 final evidence, on a GPU box from the repo root: bash profiles/tools/r05_final.sh
set -u
mkdir -p gpurun_out
timeout 900 bash profiles/tools/r05_collect.sh r05g > gpurun_out/r05g_collect.txt 2>&1
tail -25 gpurun_out/r05g_collect.txt | cut -c1-300
python bench.py > gpurun_out/r05g_bench_default.json 2>/dev/null; cut -c1-200 gpurun_out/r05g_bench_default.json
for c in 4 5 6; do python bench.py --config $c --steps 300 --warmup 100 --no-cpu-baseline --no-kernel-events --no-em-run 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('config $c ms_per_step %.4f' % d['ms_per_step'])"; done | tee gpurun_out/r05g_configs.txt
python profiles/tools/nb_step_time.py 2>&1 | tail -3 | tee gpurun_out/r05g_nb.txt
bash profiles/tools/r05_cfg_stats.sh > gpurun_out/r05g_cfg_stats.txt 2>&1
