#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02ad
mkdir -p "$out"
timeout 900 python -m pytest tests/test_din_gpu.py tests/test_din_tower_models_gpu.py tests/test_feat_api_gpu.py tests/test_zz_din_device_loader_gpu.py tests/test_retrain_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
timeout 300 python scripts/model_suite.py din > "$out/din.log" 2>&1
tail -n 3 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
grep "^din" "$out/din.log" >> "$out/summary.txt"
cat "$out/summary.txt"
