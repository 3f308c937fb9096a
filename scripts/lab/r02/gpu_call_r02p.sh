#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02p
mkdir -p "$out"
for nt in 0 1 2 3; do
  LIBRECO_ROWS_NT=$nt timeout 200 python scripts/fused_kbench.py adam 8 2>&1 | grep "^adam" | sed "s/^/nt=$nt /" >> "$out/summary.txt"
done
cat "$out/summary.txt"
