#!/bin/bash
# builds build/lab/libreco_tk_<tag>.so = the product library with csrc/score_topk.hip compiled with the given -D flags
# usage: topk_build.sh tag1 "-DLR_TK_LAB_NONORM=1" tag2 "..." ...
set -e
cd "$(dirname "$0")/../../.."
mkdir -p build/lab
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude $flags -c librecommender_amd/csrc/score_topk.hip -o build/lab/tk_$tag.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/hip/*.o | grep -v /score_topk.o) build/lab/tk_$tag.o -o build/lab/libreco_tk_$tag.so && echo "built tk_$tag" ) &
done
wait
