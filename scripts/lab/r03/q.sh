#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03q
mkdir -p "$out"
for i in 1 2 3; do
FIT_NF=20 FIT_BENCH_ONLY="device loader, hipGraph" timeout 600 python -u -W ignore scripts/fit_bench.py > "$out/run_$i.txt" 2>&1
echo "run $i rc=$? $(grep -E 'epoch|fault' "$out/run_$i.txt" | tail -1 | cut -c1-160)" >> "$out/summary.txt"
done
timeout 900 python -m pytest tests/test_graph_fit_gpu.py tests/test_din_fused_gpu.py tests/test_deepfm_fused_gpu.py tests/test_lightgcn_gpu.py tests/test_ops_gpu.py -m gpu -q --timeout 900 > "$out/t.log" 2>&1
echo "tests rc=$? $(tail -n 1 "$out/t.log" | cut -c1-160)" >> "$out/summary.txt"
cat "$out/summary.txt"
