#!/bin/bash
cd "$(dirname "$0")/../.."
python profiles/tools/fuzz_modes.py 80000 ${1:-8000} 2>&1 | tail -4
FUZZ_OPTIONS=1 python profiles/tools/fuzz_cli.py 30000 ${2:-1500} 2>&1 | grep -v "^seed.*IDENTICAL" | tail -40
