#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03s
mkdir -p "$out"
timeout 900 python -m pytest tests/test_din_gpu.py tests/test_din_fused_gpu.py tests/test_graph_fit_gpu.py tests/test_deepfm_fused_gpu.py tests/test_fullsize_cfg345_gpu.py -m gpu -q -x > "$out/t.log" 2>&1; echo "tests rc=$?"; tail -3 "$out/t.log"
timeout 600 python bench.py --workload din --steps 50 --warmup 10 --no-cpu-baseline > "$out/bench_din.json" 2> "$out/bench_din.err"; echo "din rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r03s/bench_din.json"):
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"], d["value"]); print({k: round(v["mean_ms"], 4) for k, v in d["kernels"].items()})
PY
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recommend > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "deepfm rc=$?"
grep -o '"ms_per_step": [0-9.]*' "$out/bench_default.json" | head -2
