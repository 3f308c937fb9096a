#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02h
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_softmax_ce_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "sce tests rc=$?" >> "$out/summary.txt"
timeout 300 python scripts/sce_bench.py 65536 128 4 > "$out/sce_bench.log" 2>&1; echo "sce bench rc=$?" >> "$out/summary.txt"
timeout 300 python scripts/sce_bench.py 8192 64 4 >> "$out/sce_bench.log" 2>&1
timeout 600 python -m pytest tests/test_retrain_gpu.py tests/test_din_tower_models_gpu.py tests/test_feat_api_gpu.py -m gpu -q -x --timeout 600 -k "tower or Tower or two" > "$out/tests_tt.log" 2>&1; echo "twotower tests rc=$?" >> "$out/summary.txt"
timeout 300 python scripts/model_suite.py twotower > "$out/model_tt.log" 2>&1; echo "model suite rc=$?" >> "$out/summary.txt"
tail -n 15 "$out/tests.log" | cut -c1-400 >> "$out/summary.txt"
cat "$out/sce_bench.log" >> "$out/summary.txt"
tail -n 5 "$out/tests_tt.log" | cut -c1-300 >> "$out/summary.txt"
tail -n 3 "$out/model_tt.log" >> "$out/summary.txt"
cat "$out/summary.txt"
