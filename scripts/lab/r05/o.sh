#!/bin/bash
# r05o: split-bf16 softmax-CE variants (skew of the second wave per SIMD, ring depth) + PMC of the default build
out=gpurun_out/r05o; mkdir -p $out
export SCE_BENCH_ARITHS=split_bf16
for v in default skew24 skew48 nb4 nb4skew; do
  lib=$PWD/build/lab/libreco_sce_$v.so; [ $v = default ] && lib=$PWD/librecommender_amd/lib/liblibreco_hip.so
  echo "== $v"
  LIBRECO_HIP_LIB=$lib timeout 120 python scripts/sce_bench.py 65536 128 3 2>&1 | grep -v amdgpu.ids
done | tee $out/variants.log
bash scripts/pmc_cmd.sh r05sceA "python scripts/sce_bench.py 65536 128 1" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" 2>&1 | grep -E "softmax_ce" | cut -c1-600 | tee $out/pmc.log
