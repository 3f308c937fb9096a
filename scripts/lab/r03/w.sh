#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03w
mkdir -p "$out"
timeout 900 python -m pytest tests/test_graph_fit_gpu.py tests/test_din_fused_gpu.py tests/test_zz_din_device_loader_gpu.py tests/test_device_loader_gpu.py tests/test_api_gpu.py tests/test_feat_api_gpu.py -m gpu -q -x > "$out/t.log" 2>&1; echo "tests rc=$?"; tail -3 "$out/t.log"
for i in 1 2; do
timeout 600 python -u -W ignore scripts/din_fit_bench.py > "$out/din_fit_$i.txt" 2>&1; echo "din rc=$?"
grep -E "epoch|tables" "$out/din_fit_$i.txt" | cut -c1-220
done
FIT_BENCH_ONLY="device loader, hipGraph" timeout 900 python -u -W ignore scripts/fit_bench.py > "$out/fit_42.txt" 2>&1; echo "fit42 rc=$?"
grep -E "epoch" "$out/fit_42.txt" | cut -c1-220
FIT_BENCH_ONLY="device loader, hipGraph" FIT_NF=100 FIT_N=1000000 timeout 900 python -u -W ignore scripts/fit_bench.py > "$out/fit_202.txt" 2>&1; echo "fit202 rc=$?"
grep -E "epoch" "$out/fit_202.txt" | cut -c1-220
