#!/bin/bash
# round-5 fuzz batch (sub-passes, cached row blocks, host-summed totals, independent summary workers in the command line)
python profiles/tools/fuzz_modes.py 40000 ${1:-3000} 2>&1 | tail -4
FUZZ_OPTIONS=1 python profiles/tools/fuzz_cli.py 14000 ${2:-500} 2>&1 | grep -v "^seed.*IDENTICAL" | tail -30
