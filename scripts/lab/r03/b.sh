#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03b
mkdir -p "$out"
run() {
  local name=$1 to=$2; shift 2
  timeout "$to" python -m pytest "$@" -m gpu -q --timeout 900 > "$out/t_$name.log" 2>&1
  echo "$name rc=$? $(tail -n 1 "$out/t_$name.log" | cut -c1-160)" >> "$out/summary.txt"
}
run dinfused 600 tests/test_din_fused_gpu.py
run graphfit 600 tests/test_graph_fit_gpu.py
run fs_din 900 tests/test_fullsize_cfg345_gpu.py -k "din_cfg3"
timeout 600 python bench.py --workload din --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_din.json" 2> "$out/bench_din.err"
echo "bench din rc=$? $(head -c 400 "$out/bench_din.json")" >> "$out/summary.txt"
timeout 600 python bench.py --workload din --steps 20 --warmup 5 --no-cpu-baseline --no-graph > "$out/bench_din_eager.json" 2> "$out/bench_din_eager.err"
echo "bench din eager rc=$? $(head -c 400 "$out/bench_din_eager.json")" >> "$out/summary.txt"
cat "$out/summary.txt"
