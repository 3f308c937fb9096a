#!/bin/bash
# r04g: wide forward kernel with / without the per-gap spreading of the non-MFMA instructions (interleaved A/B on one box)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04g
mkdir -p "$out"
for r in 1 2; do
for n in 0 32; do
  if [ $n = 0 ]; then L=$PWD/librecommender_amd/lib/liblibreco_hip.so; else L=$PWD/build/lab/libreco_ab$n.so; fi
  LIBRECO_HIP_LIB=$L timeout 200 python scripts/fused_kbench.py fwd 10 --tile 64 > "$out/ab_${n}_$r.txt" 2>&1; echo "ablate $n: $(tail -1 $out/ab_${n}_$r.txt | cut -c1-60)"
done
done
LIBRECO_HIP_LIB=$PWD/build/lab/libreco_ab47.so timeout 200 python scripts/fused_kbench.py fwd 10 --tile 64 > "$out/ab_47.txt" 2>&1; echo "ablate 47: $(tail -1 $out/ab_47.txt | cut -c1-60)"
