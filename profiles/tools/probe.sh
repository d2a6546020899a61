#!/bin/bash
# bash profiles/tools/probe.sh "<EXTRA 1>" "<EXTRA 2>" ... : rebuild with each set of flags on the box, print per-kernel times
cd "$(dirname "$0")/../.."
for X in "$@"; do
  touch flagger_amd/csrc/hf_estep.hip
  make -C flagger_amd/csrc EXTRA="$X" > /dev/null 2>&1 || { echo "build failed for [$X]"; continue; }
  python profiles/tools/kernel_probe.py "[$X]" 2>&1 | grep -v amdgpu.ids
done
touch flagger_amd/csrc/hf_estep.hip; make -C flagger_amd/csrc > /dev/null 2>&1
