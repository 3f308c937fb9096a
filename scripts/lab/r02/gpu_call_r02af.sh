#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02af
mkdir -p "$out"
timeout 600 python scripts/rec_bench.py > "$out/rec_bench.log" 2>&1; echo "rc=$?" >> "$out/summary.txt"
tail -n 4 "$out/rec_bench.log" >> "$out/summary.txt"
cat "$out/summary.txt"
