#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r03final2
mkdir -p "$out"
timeout 1500 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -1 "$out/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1; echo "smoke rc=$?"
timeout 1200 python bench.py --workload lightgcn --steps 20 --warmup 5 > "$out/bench_lightgcn.json" 2> "$out/bench_lightgcn.err"; echo "lightgcn rc=$?"
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_lg -o kt -- python $ROOT/bench.py --workload lightgcn --steps 5 --warmup 3 --no-cpu-baseline --steady-seconds 0 > $out/prof_lightgcn.log 2>&1)
f=$(find /tmp/prof_lg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats_lightgcn.csv"
grep "^{" $out/prof_lightgcn.log | tail -1 > $out/bench_lightgcn_prof.json
grep -o '"ms_per_step": [0-9.]*' "$out/bench_lightgcn.json" | head -1
