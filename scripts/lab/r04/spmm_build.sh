#!/bin/bash
# builds build/lab/libreco_sp_<LONG>_<WIDE>.so = the product library with csrc/spmm.hip compiled with -DLR_SP_LONGBLOCKS / -DLR_SP_CHUNKWIDE
set -e
cd "$(dirname "$0")/../../.."
mkdir -p build/lab
for v in "$@"; do
  L=${v%_*}; W=${v#*_}
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DLR_SP_LONGBLOCKS=$L -DLR_SP_CHUNKWIDE=$W -c librecommender_amd/csrc/spmm.hip -o build/lab/sp_$v.o &&
    hipcc -shared -fPIC --offload-arch=gfx950 $(ls build/hip/*.o | grep -v "/spmm.o") build/lab/sp_$v.o -o build/lab/libreco_sp_$v.so && echo "built sp_$v" ) &
done
wait
