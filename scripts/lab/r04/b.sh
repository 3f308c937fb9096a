#!/bin/bash
# r04b: wide forward kernel: parity (bit-exact vs the 32-sample kernel, fp64) + isolated timing, both tilings on the same box
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04b
mkdir -p "$out"
timeout 600 python -m pytest tests/test_l1_wide_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$out/pytest.log"
for t in 32 64 32 64; do
  timeout 200 python scripts/fused_kbench.py fwd 10 --tile $t > "$out/kbench_fwd_$t.txt" 2>&1; tail -1 "$out/kbench_fwd_$t.txt"
done
