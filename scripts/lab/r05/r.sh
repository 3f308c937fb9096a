#!/bin/bash
# r05r: default (8 waves, simple loop, VGPR-form MFMA) against the same without the flag and the pipelined form; f32 chain beside
out=gpurun_out/r05r; mkdir -p $out
export SCE_BENCH_ARITHS=split_bf16
for v in default noflag pipe8; do
  lib=$PWD/build/lab/libreco_sce_$v.so; [ $v = default ] && lib=$PWD/librecommender_amd/lib/liblibreco_hip.so
  echo "== $v"
  LIBRECO_HIP_LIB=$lib timeout 120 python scripts/sce_bench.py 65536 128 3 2>&1 | grep -v amdgpu.ids
done | tee $out/variants.log
SCE_BENCH_ARITHS=f32_chain timeout 120 python scripts/sce_bench.py 65536 128 3 2>&1 | grep -v amdgpu.ids | tee -a $out/variants.log
timeout 300 python -m pytest tests/test_softmax_ce_gpu.py -x -q 2>&1 | tail -3 | tee $out/pytest.log
