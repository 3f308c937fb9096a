#!/bin/bash
# r06g: PMC passes over the DIN step's attention kernels
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONPATH=$PWD
bash scripts/pmc_cmd.sh r06din "python bench.py --workload din --steps 6 --warmup 3 --no-cpu-baseline --steady-seconds 0" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_WAVES" \
  "FETCH_SIZE" "WRITE_SIZE" 2>&1 | grep -E "din_bwd_data|din_bwd_param|din_fwd" | cut -c1-700
