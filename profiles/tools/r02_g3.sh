timeout 900 python -m pytest tests/test_estep_gpu.py tests/test_cli_gpu.py -m gpu -x -q 2>&1 | tail -8
bash profiles/tools/sweep.sh "$@"
