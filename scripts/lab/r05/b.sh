#!/bin/bash
# r05b: remaining sb tests, ablations of the split-bf16 kernels (what is their time made of), PMC passes
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r05b; mkdir -p $out
timeout 600 python -m pytest tests/test_l1_split_bf16_gpu.py -q -m gpu > $out/sb_tests.log 2>&1; echo "sb tests rc=$?"; tail -5 $out/sb_tests.log | cut -c1-300
for n in 0 1 2 4 8 3; do
  lib=librecommender_amd/lib/liblibreco_hip.so; [ $n != 0 ] && lib=build/lab/libreco_sb$n.so
  echo "== ablate $n"; LR_KBENCH_QUICK=1 LIBRECO_HIP_LIB=$PWD/$lib timeout 200 python scripts/l1_sb_kbench.py 2>&1 | grep -E "split-bf16" | cut -c1-150
done
export LR_KBENCH_QUICK=1
bash scripts/pmc_cmd.sh r05sbA "python scripts/l1_sb_kbench.py" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum" 2>&1 | grep -E "l1_(fwd|wgrad|dgrad)" | cut -c1-460
