#!/bin/bash
# builds build/lab/libreco_sp_<tag>.so = the product library with csrc/spmm.hip compiled with the given -D flags (run in the build container)
# usage: spmm_build.sh tag1 "-DLR_SP_NT_STREAM=1" tag2 "-DLR_SP_NT_COLD=16384" ...
set -e
cd "$(dirname "$0")/../../.."
mkdir -p build/lab
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude $flags -c librecommender_amd/csrc/spmm.hip -o build/lab/spmm_$tag.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/hip/*.o | grep -v /spmm.o) build/lab/spmm_$tag.o -o build/lab/libreco_sp_$tag.so && echo "built sp_$tag" ) &
done
wait
