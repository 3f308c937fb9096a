#!/bin/bash
# r04h: parity of the wide kernels + the fused-step tests, isolated timing, then the FULL default bench line (all workloads)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04h
mkdir -p "$out"
timeout 900 python -m pytest tests/test_l1_wide_gpu.py tests/test_deepfm_fused_gpu.py tests/test_fullsize_parity_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$out/pytest.log"
for t in 32 64; do
  timeout 200 python scripts/fused_kbench.py l1 10 --tile $t > "$out/kbench_l1_$t.txt" 2>&1; tail -3 "$out/kbench_l1_$t.txt" | cut -c1-48
done
/usr/bin/time -v timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_default.json" 2> "$out/bench_default.err"; echo "bench rc=$?"
grep -E "Elapsed|Maximum resident" "$out/bench_default.err"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04h/bench_default.json").read().strip().splitlines()[-1])
print("deepfm ms", d["ms_per_step"], "loss", d["config"]["final_loss"], "steady", d.get("steady_state"))
print("kernels", {k:v["mean_ms"] for k,v in d["kernels"].items()})
print("recommend", {k:d["recommend"].get(k) for k in ("value","ms_per_pass","error")}, d["recommend"].get("roofline",{}).get("frac"))
for k,v in d.get("workloads",{}).items():
    print(k, {x:v.get(x) for x in ("ms_per_step","value","error")}, v.get("roofline",{}).get("frac"), v.get("config",{}).get("final_loss"), (v.get("cpu_baseline") or {}).get("value"))
print("cpu", d.get("cpu_baseline",{}).get("value"), "dense_adam", d.get("dense_adam"))
PY
