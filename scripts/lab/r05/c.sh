#!/bin/bash
# r05c: dgrad v2 (coalesced interleaved stores), fwd 64-tile x 2 per CU + conflict-free plane writes, wgrad with 8 multiplying waves:
# tests, kernel-trace timings of every variant, ablations of the automatic variants
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r05c; mkdir -p $out
timeout 600 python -m pytest tests/test_l1_split_bf16_gpu.py -q -x -m gpu > $out/sb_tests.log 2>&1; echo "sb tests rc=$?"; tail -12 $out/sb_tests.log | cut -c1-300
bash scripts/trace_cmd.sh r05c_full "python scripts/l1_sb_kbench.py > $PWD/$out/kbench.log 2>&1" "l1_" 2>&1 | cut -c1-130
grep -E "err|equal" $out/kbench.log | cut -c1-200 | head -40
for n in 1 2 4 8 16 32 64 3; do
  echo "== ablate $n"; LR_KBENCH_QUICK=1 LIBRECO_HIP_LIB=$PWD/build/lab/libreco_sb$n.so bash scripts/trace_cmd.sh r05c_ab$n "python scripts/l1_sb_kbench.py" "_sb_kernel" 2>&1 | cut -c1-130
done
