#!/bin/bash
# r04a: PMC pass that explains the l1_* MFMA gap (SQ wait / active split, LDS, TA/TCP stalls), softmax_ce beside it as the 0.81 reference
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04a
mkdir -p "$out"
timeout 200 python scripts/fused_kbench.py l1 10 > "$out/kbench_l1.txt" 2>&1; cat "$out/kbench_l1.txt" | tail -5
timeout 200 python scripts/sce_bench.py 65536 128 3 > "$out/sce.txt" 2>&1; tail -4 "$out/sce.txt"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
P3="TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES"
P4="TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_TA_DATA_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES"
P5="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_SMEM"
bash scripts/pmc_cmd.sh r04l1 "python scripts/fused_kbench.py l1 3" "$P1" "$P2" "$P3" "$P4" "$P5" > "$out/pmc_l1.log" 2>&1
bash scripts/pmc_cmd.sh r04sce "python scripts/sce_bench.py 65536 128 2" "$P1" "$P2" "$P3" "$P4" "$P5" > "$out/pmc_sce.log" 2>&1
grep -E "l1_|softmax" "$out/pmc_l1.log" "$out/pmc_sce.log" | cut -c1-600
