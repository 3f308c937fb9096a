#!/bin/bash
# r05ae: are the split-bf16 first-layer kernels power-bound?  the same launches on zero operands (kernel trace durations)
out=gpurun_out/r05ae; mkdir -p $out
LR_KBENCH_QUICK=1 bash scripts/trace_cmd.sh r05ae_real "python scripts/l1_sb_kbench.py" "l1_" 2>&1 | grep -E "sb_kernel" | tee $out/real.log
LR_KBENCH_QUICK=1 LR_KBENCH_ZEROS=1 bash scripts/trace_cmd.sh r05ae_zero "python scripts/l1_sb_kbench.py" "l1_" 2>&1 | grep -E "sb_kernel" | tee $out/zeros.log
