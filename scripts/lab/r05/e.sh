#!/bin/bash
# r05e: L2 touches ahead (weight planes, gz planes, id lines) in the three split-bf16 kernels: tests, timings, with / without touches
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r05e; mkdir -p $out
timeout 300 python -m pytest tests/test_l1_split_bf16_gpu.py -q -x -m gpu > $out/sb_tests.log 2>&1; echo "sb tests rc=$?"; tail -8 $out/sb_tests.log | cut -c1-300
export TRACE_TIMEOUT=90
bash scripts/trace_cmd.sh r05e_full "python scripts/l1_sb_kbench.py > $PWD/$out/kbench.log 2>&1" "_sb_kernel" 2>&1 | cut -c1-130
grep -E "split-bf16" $out/kbench.log | cut -c1-150
for n in 0 128 2 130; do
  lib=librecommender_amd/lib/liblibreco_hip.so; [ $n != 0 ] && lib=build/lab/libreco_sb$n.so
  echo "== ablate $n"; LR_KBENCH_QUICK=1 LIBRECO_HIP_LIB=$PWD/$lib bash scripts/trace_cmd.sh r05e_ab$n "python scripts/l1_sb_kbench.py" "_sb_kernel" 2>&1 | cut -c1-130
done
