#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03final3
mkdir -p "$out"
timeout 1500 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -1 "$out/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "default rc=$?"
grep -o '"ms_per_step": [0-9.]*' "$out/bench_default.json" | head -1
