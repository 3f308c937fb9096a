#!/bin/bash
# r05d: the whole GPU suite + smoke + the DeepFM bench leg with the split-bf16 first layer as the default
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r05d; mkdir -p $out
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q --timeout 300 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -30 $out/pytest_gpu.log | cut -c1-250
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.log | cut -c1-200
t0=$(date +%s)
timeout 400 python bench.py --steps 20 --warmup 5 --no-workloads --no-recommend --no-cpu-baseline --no-dense-adam-line > $out/bench_deepfm.json 2> $out/bench_deepfm.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
try:
    r=json.loads(open("gpurun_out/r05d/bench_deepfm.json").read().strip().splitlines()[-1])
    print({k:r.get(k) for k in ("value","ms_per_step","dtype","steady_ms_per_step","f32_chain_ms_per_step","sum_kernel_ms")})
    for k,v in r["kernels"].items(): print(" ",k,v)
    print(r["config"]["final_loss"], r.get("f32_chain"))
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r05d/bench_deepfm.err").read()[-2000:])
PY
