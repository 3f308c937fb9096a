#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03y
mkdir -p "$out"
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recommend > "$out/bench_default_$i.json" 2> "$out/bench_default_$i.err"; echo "deepfm rc=$?"
python - "$out/bench_default_$i.json" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print(d["ms_per_step"], d["value"], d.get("steady_state")); print({k: round(v["mean_ms"], 4) for k, v in d["kernels"].items() if v["mean_ms"] > 0.05})
PY
done
