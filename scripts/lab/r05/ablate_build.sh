#!/bin/bash
# builds build/lab/libreco_sb<N>.so = the product library with csrc/deepfm_l1_sb.hip compiled with -DLR_SB_ABLATE=N (run in the build container)
set -e
cd "$(dirname "$0")/../../.."
mkdir -p build/lab
for n in "$@"; do
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -DLR_SB_ABLATE=$n -c librecommender_amd/csrc/deepfm_l1_sb.hip -o build/lab/l1_sb$n.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/hip/*.o | grep -v deepfm_l1_sb.o) build/lab/l1_sb$n.o -o build/lab/libreco_sb$n.so && echo "built sb$n" ) &
done
wait
