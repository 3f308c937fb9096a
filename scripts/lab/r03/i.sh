#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03i
mkdir -p "$out"
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --timeout 1200 ) > "$out/pytest_gpu.log" 2>&1
echo "pytest -m gpu rc=$? $(grep -E 'passed|failed' "$out/pytest_gpu.log" | tail -1 | cut -c1-160)" >> "$out/summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke rc=$?" >> "$out/summary.txt"
for w in din twotower lightgcn; do
  ( time timeout 900 python bench.py --workload $w --steps 20 --warmup 5 ) > "$out/bench_$w.json" 2> "$out/bench_$w.err"
  echo "bench $w rc=$? $(head -c 160 "$out/bench_$w.json")" >> "$out/summary.txt"
done
cat "$out/summary.txt"
