#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03final
mkdir -p "$out"
timeout 1500 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -2 "$out/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$out/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$out/smoke.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "default rc=$?"
for w in din twotower lightgcn; do
  timeout 1200 python bench.py --workload $w --steps 20 --warmup 5 > "$out/bench_$w.json" 2> "$out/bench_$w.err"; echo "$w rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03final/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"))
PY
