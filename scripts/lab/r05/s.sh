#!/bin/bash
# r05s: what bounds the split-bf16 softmax-CE sweeps?  zero operands (power), no LDS reads in either contraction (LDS)
out=gpurun_out/r05s; mkdir -p $out
export SCE_BENCH_ARITHS=split_bf16
( echo "== default, operands of zeros"; SCE_BENCH_ZEROS=1 timeout 120 python scripts/sce_bench.py 65536 128 3 2>&1 | grep -v amdgpu.ids
for v in abl1 abl2 abl3; do
  echo "== $v"
  LIBRECO_HIP_LIB=$PWD/build/lab/libreco_sce_$v.so timeout 120 python scripts/sce_bench.py 65536 128 3 2>&1 | grep -v amdgpu.ids
done ) | tee $out/ablations.log
