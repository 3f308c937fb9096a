#!/bin/bash
# r05q: pipelined split-bf16 softmax-CE (4 waves, manual MFMA/VALU interleave): parity, variants, TwoTower line
out=gpurun_out/r05q; mkdir -p $out
timeout 600 python -m pytest tests/test_softmax_ce_gpu.py -x -q -s 2>&1 | tail -8 | tee $out/pytest.log
if ! grep -q " passed" $out/pytest.log || grep -q "failed" $out/pytest.log; then
  echo "== no-TR variant"
  LIBRECO_HIP_LIB=$PWD/build/lab/libreco_sce_notr.so timeout 600 python -m pytest tests/test_softmax_ce_gpu.py -x -q 2>&1 | tail -8 | tee $out/pytest_notr.log
fi
export SCE_BENCH_ARITHS=split_bf16
for v in default w8 nopipe; do
  lib=$PWD/build/lab/libreco_sce_$v.so; [ $v = default ] && lib=$PWD/librecommender_amd/lib/liblibreco_hip.so
  echo "== $v"
  LIBRECO_HIP_LIB=$lib timeout 120 python scripts/sce_bench.py 65536 128 3 2>&1 | grep -v amdgpu.ids
done | tee $out/variants.log
bash scripts/pmc_cmd.sh r05sceB "python scripts/sce_bench.py 65536 128 1" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" 2>&1 | grep -E "softmax_ce" | cut -c1-600 | tee $out/pmc.log
