#!/bin/bash
# r04n: round-end validation: all GPU tests, smoke, the driver's bench command, rocprofv3 kernel trace + PMC traffic of the same workload,
# the row-sharded step at world size 1, the 100 M-item scoring pass's fabric traffic
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r04n
mkdir -p "$out"
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -1 "$out/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1; echo "smoke rc=$?"
t0=$(date +%s)
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "default rc=$? wall=$(( $(date +%s) - t0 )) s"
grep -o '"ms_per_step": [0-9.]*' "$out/bench_default.json" | head -5 | tr '\n' ' '; echo
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_r04 -o kt -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line --steady-seconds 0 > $out/prof_deepfm.log 2>&1)
f=$(find /tmp/prof_r04 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats_deepfm.csv" && head -12 "$out/kernel_stats_deepfm.csv" | cut -c1-150
grep -o '"ms_per_step": [0-9.]*' "$out/prof_deepfm.log" | head -1
bash scripts/pmc_cmd.sh r04traffic "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line --steady-seconds 0 --no-graph" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" > "$out/pmc_traffic.log" 2>&1
grep -E "l1_|rows_adam|field_stats" "$out/pmc_traffic.log" | cut -c1-330
bash scripts/pmc_cmd.sh r04topk "python scripts/score_topk_traffic.py --once" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" > "$out/pmc_topk.log" 2>&1
grep -E "score_topk|merge" "$out/pmc_topk.log" | cut -c1-330
timeout 300 python bench.py --force-sharded --steps 20 --warmup 5 --no-cpu-baseline --no-recommend --no-dense-adam-line --steady-seconds 0 > "$out/force_sharded.json" 2> "$out/force_sharded.err"; echo "force-sharded rc=$?"; grep -o '"ms_per_step": [0-9.]*' "$out/force_sharded.json" | head -1
