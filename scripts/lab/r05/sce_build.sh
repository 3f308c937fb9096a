#!/bin/bash
# builds build/lab/libreco_sce_<tag>.so = the product library with csrc/softmax_ce.hip compiled with extra -D flags
# usage: sce_build.sh tag "-DLR_SCE_SKEW=24 ..." [tag flags ...]
set -e
cd "$(dirname "$0")/../../.."
mkdir -p build/lab
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c librecommender_amd/csrc/softmax_ce.hip -o build/lab/sce_$tag.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/hip/*.o | grep -v softmax_ce.o) build/lab/sce_$tag.o -o build/lab/libreco_sce_$tag.so && echo "built $tag" ) &
done
wait
