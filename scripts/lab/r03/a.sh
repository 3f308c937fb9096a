#!/bin/bash
# r03 call a: new fused DIN step, graph-stream replays, device Laplacian, full-size cfg 3/4/5 tests, bench workloads
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03a
mkdir -p "$out"
{ free -g | head -2; nproc; } > "$out/host.txt" 2>&1
timeout 120 scripts/probes/graph_null_stream_probe > "$out/graph_probe.txt" 2>&1; echo "probe rc=$?" >> "$out/summary.txt"
run() {  # name, timeout, files...
  local name=$1 to=$2; shift 2
  timeout "$to" python -m pytest "$@" -m gpu -q --timeout 900 -x > "$out/t_$name.log" 2>&1
  echo "$name rc=$? $(tail -n 1 "$out/t_$name.log" | cut -c1-160)" >> "$out/summary.txt"
}
run dinfused 600 tests/test_din_fused_gpu.py
run graphfit 600 tests/test_graph_fit_gpu.py
run lap 300 tests/test_fullsize_cfg345_gpu.py -k "device_laplacian_matches"
run fs_din 900 tests/test_fullsize_cfg345_gpu.py -k "din_cfg3"
run fs_tt 600 tests/test_fullsize_cfg345_gpu.py -k "twotower_cfg4"
run fs_lg 600 tests/test_fullsize_cfg345_gpu.py -k "lightgcn_cfg5"
run existing 1200 tests/test_deepfm_fused_gpu.py tests/test_din_gpu.py tests/test_din_tower_models_gpu.py tests/test_lightgcn_gpu.py tests/test_zz_din_device_loader_gpu.py tests/test_device_loader_gpu.py tests/test_feat_api_gpu.py
for w in din twotower lightgcn; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_$w.json" 2> "$out/bench_$w.err"
  echo "bench $w rc=$? $(head -c 300 "$out/bench_$w.json")" >> "$out/summary.txt"
done
cat "$out/summary.txt"
