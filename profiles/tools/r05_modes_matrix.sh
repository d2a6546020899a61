#!/bin/bash
# the GPU parity suites once more under forced non-default modes of the round-5 library (the tests that assert a mode's own default are deselected)
DESEL="not cached_row_blocks and not sub_passes_through and not static_guard and not host_summed and not hand_off_time_out and not one_launch"
for env in "HF_SUBPASSES=2" "HF_SUBPASSES=3 HF_SEG_CACHED_STEPS=8" "HF_TOTAL=device HF_SEG_CACHED_STEPS=3" "HF_SEG_LAUNCHES=2 HF_SUBPASSES=2" "HF_STATS=chunks HF_SUBPASSES=2"; do
  echo "== $env"
  env $env python -m pytest tests/test_estep_gpu.py tests/test_multi_gpu.py tests/test_shim_gpu.py -q -m gpu -k "$DESEL and not full_size and not cfg4_at_full" 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -6
done
