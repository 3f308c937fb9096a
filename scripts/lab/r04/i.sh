#!/bin/bash
# r04i: the FULL default bench line (all workloads), wall time recorded
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04i
mkdir -p "$out"
t0=$(date +%s)
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
tail -5 "$out/bench_default.err"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04i/bench_default.json").read().strip().splitlines()[-1])
print("deepfm ms", d["ms_per_step"], "loss", d["config"]["final_loss"], "steady", d.get("steady_state"))
print("kernels", {k:v["mean_ms"] for k,v in d["kernels"].items()})
print("recommend", {k:d["recommend"].get(k) for k in ("value","ms_per_pass","error")}, d["recommend"].get("roofline",{}).get("frac"))
for k,v in d.get("workloads",{}).items():
    print(k, {x:v.get(x) for x in ("ms_per_step","value","error")}, v.get("roofline",{}).get("frac"), v.get("config",{}).get("final_loss"), (v.get("cpu_baseline") or {}).get("value"))
print("cpu", d.get("cpu_baseline",{}).get("value"), "dense_adam", d.get("dense_adam"))
PY
