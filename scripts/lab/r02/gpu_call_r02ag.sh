#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02ag
mkdir -p "$out"
timeout 1200 python -m pytest tests/test_fm_models_gpu.py tests/test_api_gpu.py tests/test_retrain_gpu.py tests/test_device_loader_gpu.py tests/test_feat_api_gpu.py tests/test_oov_values_gpu.py -m gpu -q --timeout 600 > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
tail -n 12 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
cat "$out/summary.txt"
