#!/bin/bash
# Build-switch variants of the library for same-box A/B runs, built HERE (hipcc cross-compiles) so that no GPU minute goes into compiling:
#   bash profiles/tools/build_variants.sh "name=-DHF_SEG_RCP=0" "old=-DHF_SEG_RCP=0 -DHF_SEG_VECSUF=0" ...
# -> flagger_amd/csrc/variants/libhmmflagger_hip.<name>.so (git-ignored, travels with gpurun); select with HF_LIBRARY_VARIANT=<name>.
set -u
cd "$(dirname "$0")/../../flagger_amd/csrc"
make > /dev/null 2>&1 || { echo "default build failed"; exit 1; }
mkdir -p variants
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -I../../include"
for spec in "$@"; do
  name=${spec%%=*}; extra=${spec#*=}
  ( /opt/rocm/bin/hipcc $HIPFLAGS $extra -c hf_estep.hip -o variants/hf_estep.$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC variants/hf_estep.$name.o hf_multi.o hf_model.o hf_io.o hf_summary.o -o variants/libhmmflagger_hip.$name.so -lrccl -lz -lpthread &&
    rm -f variants/hf_estep.$name.o && echo "built $name [$extra]" || echo "FAILED $name" ) &
done
wait
