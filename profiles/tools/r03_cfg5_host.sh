HF_HOST_TRACE=1 python bench.py --config 5 --no-cpu-baseline --steps 6 --warmup 2 2>&1 | grep -v "^\[hf_create\]" | head -60
