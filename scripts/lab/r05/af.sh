#!/bin/bash
out=gpurun_out/r05af; mkdir -p $out
export SCE_BENCH_ARITHS=split_bf16
for v in default pad32 pad48 pad64; do
  lib=$PWD/build/lab/libreco_sce_$v.so; [ $v = default ] && lib=$PWD/librecommender_amd/lib/liblibreco_hip.so
  echo "== $v"
  LIBRECO_HIP_LIB=$lib timeout 120 python scripts/sce_bench.py 65536 128 3 2>&1 | grep -v amdgpu.ids
done | tee $out/variants.log
