#!/bin/bash
# last call of the round: default line on the final tree (with the re-collected traffic figures) + cfg 5 row-partitioned at world size 1 on the 400 M-nnz graph
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03last
mkdir -p "$out"
timeout 170 python bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "default rc=$?"
grep -o '"ms_per_step": [0-9.]*' "$out/bench_default.json" | head -1
timeout 150 python bench.py --workload lightgcn --force-sharded --steps 5 --warmup 2 --no-cpu-baseline --steady-seconds 0 > "$out/lg_fs.json" 2> "$out/lg_fs.err"; echo "lg fs rc=$?"
grep -o '"ms_per_step": [0-9.]*' "$out/lg_fs.json" | head -1
