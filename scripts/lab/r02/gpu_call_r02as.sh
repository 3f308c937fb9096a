#!/bin/bash
# r02as: pair_mlp at 1,024 users x 1 M items: timing + PMC (MFMA issue share, LDS conflicts)
set -x
mkdir -p gpurun_out
timeout 300 python scripts/kern_suite.py pair_mlp > gpurun_out/r02as_pair_mlp.txt 2>&1
tail -2 gpurun_out/r02as_pair_mlp.txt
bash scripts/pmc_cmd.sh pairmlp "python scripts/kern_suite.py pair_mlp" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE" > gpurun_out/r02as_pmc.txt 2>&1
tail -6 gpurun_out/r02as_pmc.txt
