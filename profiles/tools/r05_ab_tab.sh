set -x
python -m pytest tests/test_estep_gpu.py -x -q -m gpu 2>&1 | tail -8
for v in 0 1; do for i in 1 2; do HF_TAB_FUSED=$v python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('TAB_FUSED=$v ms_per_step', d['ms_per_step'])"; done; done
for w in 256 512 2048 3072; do HF_TAB_WGS=$w python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{\"metric\"')][-1]); print('TAB_WGS=$w ms_per_step', d['ms_per_step'])"; done
