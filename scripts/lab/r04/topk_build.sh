#!/bin/bash
# builds build/lab/libreco_tk_<ROWS>_<SLACK>.so = the product library with csrc/score_topk.hip compiled with -DLR_TOPK_WIN_ROWS / -DLR_TOPK_WIN_SLACK
set -e
cd "$(dirname "$0")/../../.."
mkdir -p build/lab
for v in "$@"; do
  R=${v%_*}; S=${v#*_}
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DLR_TOPK_WIN_ROWS=$R -DLR_TOPK_WIN_SLACK=$S -c librecommender_amd/csrc/score_topk.hip -o build/lab/tk_$v.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/hip/*.o | grep -v "/score_topk.o") build/lab/tk_$v.o -o build/lab/libreco_tk_$v.so && echo "built tk_$v" ) &
done
wait
