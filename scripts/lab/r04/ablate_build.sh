#!/bin/bash
# builds build/lab/libreco_ab<N>.so = the product library with csrc/deepfm_l1.hip compiled with -DLR_L1_ABLATE=N (run in the build container)
set -e
cd "$(dirname "$0")/../../.."
mkdir -p build/lab
for n in "$@"; do
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DLR_L1_ABLATE=$n -c librecommender_amd/csrc/deepfm_l1.hip -o build/lab/l1_ab$n.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/hip/*.o | grep -v deepfm_l1.o) build/lab/l1_ab$n.o -o build/lab/libreco_ab$n.so && echo "built ab$n" ) &
done
wait
