#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03c
mkdir -p "$out"
run() {
  local name=$1 to=$2; shift 2
  timeout "$to" python -m pytest "$@" -m gpu -q --timeout 900 > "$out/t_$name.log" 2>&1
  echo "$name rc=$? $(tail -n 1 "$out/t_$name.log" | cut -c1-160)" >> "$out/summary.txt"
}
run dinfused 600 tests/test_din_fused_gpu.py tests/test_graph_fit_gpu.py
run fs_din 900 tests/test_fullsize_cfg345_gpu.py -k "din_cfg3"
run scatter_users 900 tests/test_ops_gpu.py tests/test_lightgcn_gpu.py tests/test_din_tower_models_gpu.py tests/test_sharded_gpu.py tests/test_zz_ngcf_gpu.py tests/test_youtube_retrieval_gpu.py
for w in din twotower lightgcn; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-recommend > "$out/bench_$w.json" 2> "$out/bench_$w.err"
  echo "bench $w rc=$? $(head -c 200 "$out/bench_$w.json")" >> "$out/summary.txt"
done
cat "$out/summary.txt"
