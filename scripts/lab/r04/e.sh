#!/bin/bash
# r04e: all three wide kernels: parity + isolated timing against the 32-sample kernels on the same box
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04e
mkdir -p "$out"
timeout 600 python -m pytest tests/test_l1_wide_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$out/pytest.log"
for t in 32 64 32 64; do
  timeout 200 python scripts/fused_kbench.py l1 10 --tile $t > "$out/kbench_l1_$t.txt" 2>&1; tail -3 "$out/kbench_l1_$t.txt" | cut -c1-48,150-
done
