#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02x
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_device_loader_gpu.py tests/test_zz_din_device_loader_gpu.py -m gpu -q --timeout 600 > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
tail -n 30 "$out/tests.log" | cut -c1-400 >> "$out/summary.txt"
cat "$out/summary.txt"
