#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03d
mkdir -p "$out"
run() {
  local name=$1 to=$2; shift 2
  timeout "$to" python -m pytest "$@" -m gpu -q --timeout 900 > "$out/t_$name.log" 2>&1
  echo "$name rc=$? $(tail -n 1 "$out/t_$name.log" | cut -c1-160)" >> "$out/summary.txt"
}
run ops 900 tests/test_ops_gpu.py tests/test_din_fused_gpu.py
run sharded 1200 tests/test_sharded_gpu.py
run cfg1 900 tests/test_cfg1_movielens_gpu.py
run loaders 900 tests/test_device_loader_gpu.py tests/test_youtube_retrieval_gpu.py tests/test_api_gpu.py
timeout 900 python bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"
echo "bench default rc=$? $(head -c 300 "$out/bench_default.json")" >> "$out/summary.txt"
timeout 600 python bench.py --workload twotower --force-sharded --steps 5 --warmup 2 > "$out/bench_tt_sharded.json" 2> "$out/bench_tt_sharded.err"
echo "bench tt sharded rc=$? $(head -c 300 "$out/bench_tt_sharded.json")" >> "$out/summary.txt"
cat "$out/summary.txt"
