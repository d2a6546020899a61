#!/bin/bash
# round-6 final evidence, on a GPU box from the repo root: bash profiles/tools/r05_final.sh
set -u
mkdir -p gpurun_out
timeout 900 bash profiles/tools/r06_collect.sh r06i > gpurun_out/r06i_collect.txt 2>&1
tail -25 gpurun_out/r06i_collect.txt | cut -c1-300
python bench.py > gpurun_out/r06i_bench_default.json 2>/dev/null; cut -c1-200 gpurun_out/r06i_bench_default.json
for c in 4 5 6; do python bench.py --config $c --steps 300 --warmup 100 --no-cpu-baseline --no-kernel-events --no-em-run 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('config $c ms_per_step %.4f' % d['ms_per_step'])"; done | tee gpurun_out/r06i_configs.txt
python profiles/tools/nb_step_time.py 2>&1 | tail -3 | tee gpurun_out/r06i_nb.txt
bash profiles/tools/r06_cfg_stats.sh > gpurun_out/r06i_cfg_stats.txt 2>&1
# size sweep on the final build: ms per EM step behind the runtime's one-time stall (1 000 steps behind 1 500 up to x 1), sub-passes beyond
for sc in 0.125 0.25 0.5 1; do python bench.py --scale $sc --steps 1000 --warmup 1500 --no-cpu-baseline --no-em-run --event-stride 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc ms_per_step %.4f  cached blocks %s sub-passes %s' % (d['ms_per_step'], d['roofline'].get('cached_row_blocks'), d['roofline'].get('sub_passes')), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"; done | tee gpurun_out/r06i_scale.txt
for sc in 2 4 8; do python bench.py --scale $sc --steps 200 --warmup 60 --no-cpu-baseline --no-em-run --event-stride 8 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('scale $sc ms_per_step %.4f  cached blocks %s sub-passes %s' % (d['ms_per_step'], d['roofline'].get('cached_row_blocks'), d['roofline'].get('sub_passes')), {a: round(b*1e3,1) for a,b in d['roofline']['kernel_ms_all'].items()})"; done | tee -a gpurun_out/r06i_scale.txt
for sc in 0.5 0.25 0.125; do for ex in ranks chunks; do python bench.py --gpus 1 --dist-path --exchange $ex --scale $sc --steps 1000 --warmup 1500 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('dist path (one rank, RCCL all-gather) scale $sc exchange $ex ms_per_step %.4f' % d['ms_per_step'])"; done; done | tee gpurun_out/r06i_dist_path.txt
# the reference's default window length (16 kb: -x hifi / ont-r9 with no -W, ~380 k windows) in the driver's regime, with its own roofline fraction
python bench.py --scale 0.25 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/r06i_bench_quarter.json; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r06i_bench_quarter.json') if l.startswith('{\"metric\"')][-1]); r=d['roofline']
print('scale 0.25 (driver regime): ms_per_step %.4f, k_seg_fb %.1f us, %.0f GB/s = %.2f %% of HBM peak' % (d['ms_per_step'], r['kernel_ms_timed']*1e3, r['achieved'], 100*r['frac']))" | tee gpurun_out/r06i_quarter.txt
