#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03aa
mkdir -p "$out"
timeout 900 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_fullsize_parity_gpu.py tests/test_graph_fit_gpu.py -m gpu -q -x > "$out/t.log" 2>&1; echo "tests rc=$?"; tail -12 "$out/t.log" | cut -c1-300
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recommend > "$out/bench_default_$i.json" 2> "$out/bench_default_$i.err"; echo "deepfm rc=$?"
python - "$out/bench_default_$i.json" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"], d["value"], d.get("steady_state")); print({k: round(v["mean_ms"], 4) for k, v in d["kernels"].items() if v["mean_ms"] > 0.05})
PY
tail -3 "$out/bench_default_$i.err" | cut -c1-300
done
