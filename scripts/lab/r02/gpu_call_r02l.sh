#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02l
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_sharded_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "sharded tests rc=$?" >> "$out/summary.txt"
timeout 300 python bench.py --no-cpu-baseline --no-recommend --force-sharded > "$out/bench_sharded.json" 2> "$out/bench_sharded.err"; echo "bench force-sharded rc=$?" >> "$out/summary.txt"
tail -n 15 "$out/tests.log" | cut -c1-400 >> "$out/summary.txt"
cut -c1-4000 "$out/bench_sharded.json" >> "$out/summary.txt"; tail -n 4 "$out/bench_sharded.err" | cut -c1-400 >> "$out/summary.txt"
cat "$out/summary.txt"
