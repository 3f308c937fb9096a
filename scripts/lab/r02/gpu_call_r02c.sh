#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02c
mkdir -p "$out"
for A in 0 1 2 4 3 7; do
  echo "== ablate $A" >> "$out/ablate.log"
  LIBRECO_L1_ABLATE=$A timeout 200 python scripts/fused_kbench.py l1 5 2>&1 | grep -E "ms:" >> "$out/ablate.log"
done
timeout 600 python scripts/diag_fullsize.py --wgrad --oracle > "$out/diag.log" 2>&1; echo "diag rc=$?" >> "$out/summary.txt"
bash scripts/pmc_cmd.sh l1 "python scripts/fused_kbench.py l1 3" \
  "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" \
  "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" > "$out/pmc_l1.log" 2>&1
bash scripts/pmc_cmd.sh adam "python scripts/fused_kbench.py adam 3" \
  "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum SQ_WAVES SQ_WAIT_ANY SQ_WAVE_CYCLES" > "$out/pmc_adam.log" 2>&1
for f in ablate diag pmc_l1 pmc_adam; do echo "== $f"; tail -n 40 "$out/$f.log" | cut -c1-900; done >> "$out/summary.txt" 2>/dev/null
tail -n 200 "$out/summary.txt"
