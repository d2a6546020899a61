#!/bin/bash
# round-6 fuzz batch on the final build (split hf_create, per-thread key marks, wide hand-off, padded record staging, pair-decoding loader)
cd "$(dirname "$0")/../.."
python profiles/tools/fuzz_modes.py 70000 ${1:-3000} 2>&1 | tail -4
FUZZ_OPTIONS=1 python profiles/tools/fuzz_cli.py 20000 ${2:-600} 2>&1 | grep -v "^seed.*IDENTICAL" | tail -30
