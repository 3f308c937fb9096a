#!/bin/bash
# final validation + cfg 5 with 200 M DISTINCT interactions (400 M nnz): bench line, SpMM traffic (PMC), kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r03lg400
mkdir -p "$out"
timeout 600 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -1 "$out/pytest_gpu.log"
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1; echo "smoke rc=$?"
timeout 300 python bench.py --workload lightgcn --steps 20 --warmup 5 > "$out/bench_lightgcn.json" 2> "$out/bench_lightgcn.err"; echo "lightgcn rc=$?"
grep -o '"ms_per_step": [0-9.]*' "$out/bench_lightgcn.json" | head -1; grep -o '"nnz": [0-9]*' "$out/bench_lightgcn.json" | head -1
bash scripts/pmc_cmd.sh r03lg400 "python bench.py --workload lightgcn --steps 2 --warmup 1 --no-cpu-baseline --steady-seconds 0" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" > "$out/pmc_lightgcn.log" 2>&1
grep -E "spmm|adam_dense" "$out/pmc_lightgcn.log" | cut -c1-320
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_lg -o kt -- python $ROOT/bench.py --workload lightgcn --steps 5 --warmup 3 --no-cpu-baseline --steady-seconds 0 > $out/prof_lightgcn.log 2>&1)
f=$(find /tmp/prof_lg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats_lightgcn.csv"
grep "^{" $out/prof_lightgcn.log | tail -1 > $out/bench_lightgcn_prof.json
timeout 200 python bench.py --force-sharded --steps 5 --warmup 2 --no-cpu-baseline --no-recommend --no-dense-adam-line --steady-seconds 0 > "$out/fs.json" 2> "$out/fs.err"; echo "fs rc=$? stdout lines=$(wc -l < $out/fs.json)"
