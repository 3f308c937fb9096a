#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03g
mkdir -p "$out"
timeout 900 python -m pytest tests/test_din_gpu.py tests/test_din_fused_gpu.py tests/test_din_tower_models_gpu.py tests/test_feat_api_gpu.py -m gpu -q --timeout 900 > "$out/t_din.log" 2>&1
echo "din rc=$? $(tail -n 1 "$out/t_din.log" | cut -c1-160)" >> "$out/summary.txt"
timeout 600 python bench.py --workload din --steps 50 --warmup 10 --no-cpu-baseline > "$out/bench_din.json" 2> "$out/bench_din.err"
echo "bench din rc=$? $(head -c 200 "$out/bench_din.json")" >> "$out/summary.txt"
python - <<'PY' >> "$out/summary.txt"
import json
d=json.load(open("gpurun_out/r03g/bench_din.json"))
print({k:v['mean_ms'] for k,v in d['kernels'].items() if 'din' in k}, d['ms_per_step'])
PY
cat "$out/summary.txt"
