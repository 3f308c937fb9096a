#!/bin/bash
# r04 final: all GPU tests, smoke, the driver's bench command, kernel trace of the DeepFM step, the row-sharded step at world size 1
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r04final
mkdir -p "$out"
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -1 "$out/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1; echo "smoke rc=$?"
t0=$(date +%s)
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "default rc=$? wall=$(( $(date +%s) - t0 )) s"
grep -o '"ms_per_step": [0-9.]*' "$out/bench_default.json" | head -5 | tr '\n' ' '; echo
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_r04 -o kt -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line --steady-seconds 0 > $out/prof_deepfm.json 2> $out/prof_deepfm.err)
f=$(find /tmp/prof_r04 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats_deepfm.csv" && head -8 "$out/kernel_stats_deepfm.csv" | cut -c1-150
grep -o '"ms_per_step": [0-9.]*' "$out/prof_deepfm.json" | head -1
timeout 300 python bench.py --force-sharded --steps 20 --warmup 5 --no-cpu-baseline --no-recommend --no-dense-adam-line --steady-seconds 0 > "$out/force_sharded.json" 2> "$out/force_sharded.err"; echo "force-sharded rc=$?"; grep -o '"ms_per_step": [0-9.]*' "$out/force_sharded.json" | head -1
timeout 300 python bench.py --no-workloads --no-recommend --no-cpu-baseline --no-dense-adam-line --steady-seconds 0 --steps 20 --warmup 5 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
