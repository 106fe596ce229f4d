#!/bin/bash
# tools/build_variant.sh NAME "<extra hipcc flags>" -> gpupathtracer_amd/lib/libgpt_hip_NAME.so
# (an A/B build of the library with different kernel macros, e.g. -DKD_WAVES_PER_EU=5 -DKD_LDS_N=16; select it with GPT_LIB_PATH)
set -e
cd "$(dirname "$0")/../gpupathtracer_amd/csrc"
name=$1; shift
tmp=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-result -Wno-unused-value $*"
/opt/rocm/bin/hipcc $FLAGS -c -o $tmp/k.o gpt_kernels.hip
/opt/rocm/bin/hipcc $FLAGS -c -o $tmp/kb.o kdtree_build_gpu.hip &
for f in gpt_api gpt_host_io kdtree_build; do /opt/rocm/bin/hipcc $FLAGS -x hip -c -o $tmp/$f.o $f.cpp & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgpt_hip_$name.so $tmp/*.o -ldl
rm -rf $tmp
echo built libgpt_hip_$name.so
