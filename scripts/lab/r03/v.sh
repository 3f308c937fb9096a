#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03v
mkdir -p "$out"
timeout 600 python -u -W ignore scripts/din_fit_bench.py > "$out/din_fit.txt" 2>&1; echo "din rc=$?"
grep -E "epoch|tables" "$out/din_fit.txt" | cut -c1-220
timeout 900 python -u -W ignore scripts/fit_bench.py > "$out/fit_42.txt" 2>&1; echo "fit42 rc=$?"
grep -E "epoch" "$out/fit_42.txt" | cut -c1-220
FIT_NF=100 FIT_N=1000000 timeout 900 python -u -W ignore scripts/fit_bench.py > "$out/fit_202.txt" 2>&1; echo "fit202 rc=$?"
grep -E "epoch" "$out/fit_202.txt" | cut -c1-220
