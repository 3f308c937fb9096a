#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03r
mkdir -p "$out"
timeout 900 python -u -W ignore scripts/din_fit_bench.py > "$out/din_fit.txt" 2>&1; echo "rc=$?" >> "$out/din_fit.txt"
grep -vE "Warning|warn" "$out/din_fit.txt" | tail -8 | cut -c1-200
timeout 600 python -m pytest tests/test_graph_fit_gpu.py tests/test_din_fused_gpu.py -m gpu -q > "$out/t.log" 2>&1; tail -1 "$out/t.log"
