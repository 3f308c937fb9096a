#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03ad
mkdir -p "$out"
timeout 600 python -u -W ignore scripts/din_fit_bench.py > "$out/din_fit.txt" 2>&1; echo "din rc=$?"
grep -E "epoch|tables" "$out/din_fit.txt" | cut -c1-220
for nf in 20 100; do
  for tag in "device loader, eager" "device loader, hipGraph"; do
    FIT_BENCH_ONLY="$tag" FIT_NF=$nf FIT_N=$([ $nf = 100 ] && echo 1000000 || echo 2000000) timeout 900 python -u -W ignore scripts/fit_bench.py 2>&1 | grep -E "epoch" | cut -c1-200
  done
done
