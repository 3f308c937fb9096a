#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r03e
mkdir -p "$out"
export TMPDIR=/tmp
run() {
  local name=$1 to=$2; shift 2
  timeout "$to" python -m pytest "$@" -m gpu -q --timeout 900 > "$out/t_$name.log" 2>&1
  echo "$name rc=$? $(tail -n 1 "$out/t_$name.log" | cut -c1-160)" >> "$out/summary.txt"
}
run din 900 tests/test_din_fused_gpu.py tests/test_graph_fit_gpu.py tests/test_cfg1_movielens_gpu.py
timeout 600 python bench.py --workload din --steps 50 --warmup 10 --no-cpu-baseline > "$out/bench_din.json" 2> "$out/bench_din.err"
echo "bench din rc=$? $(head -c 200 "$out/bench_din.json")" >> "$out/summary.txt"
# kernel traces (rocprofv3 --kernel-trace --stats) of the din and default lines
for w in din deepfm; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $out/prof_$w -o kt -- python $ROOT/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-recommend --steady-seconds 0 > $out/prof_$w.log 2>&1)
  f=$(find $out/prof_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w $f" >> "$out/summary.txt"
  [ -n "$f" ] && head -40 "$f" > "$out/kernel_stats_$w.csv"
done
cat "$out/summary.txt"
