#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out; mkdir -p $O
HF_PARTS_TRACE=1 PROBE_N=4 python profiles/tools/r06_create_probe.py > $O/r06_parts_trace.txt 2>&1
grep -c parts $O/r06_parts_trace.txt; grep "context\|median" $O/r06_parts_trace.txt | cut -c1-200
grep "parts" $O/r06_parts_trace.txt | sed -n '4,9p' | cut -c1-1500
