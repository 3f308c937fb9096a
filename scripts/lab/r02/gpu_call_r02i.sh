#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02i
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_din_gpu.py tests/test_din_tower_models_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "din tests rc=$?" >> "$out/summary.txt"
timeout 300 python scripts/kern_suite.py din > "$out/din_mfma.log" 2>&1; echo "din bench rc=$?" >> "$out/summary.txt"
LIBRECO_DIN_SHUFFLE=1 timeout 300 python scripts/kern_suite.py din > "$out/din_shuffle.log" 2>&1
tail -n 25 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
grep -h "din" "$out/din_mfma.log" "$out/din_shuffle.log" >> "$out/summary.txt"
cat "$out/summary.txt"
