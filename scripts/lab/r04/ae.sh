#!/bin/bash
# r04ae: PMC of the experimental split-bf16 forward next to the f32 forward (scripts/l1_sb_kbench.py)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
bash scripts/pmc_cmd.sh r04sbA "python scripts/l1_sb_kbench.py" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" 2>&1 | grep -E "l1_fwd" | cut -c1-420
bash scripts/pmc_cmd.sh r04sbB "python scripts/l1_sb_kbench.py" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" 2>&1 | grep -E "l1_fwd" | cut -c1-420
bash scripts/pmc_cmd.sh r04sbC "python scripts/l1_sb_kbench.py" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" 2>&1 | grep -E "l1_fwd" | cut -c1-420
