#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02t
mkdir -p "$out"
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
timeout 300 python bench.py --no-cpu-baseline --no-recommend > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/summary.txt"
timeout 300 python bench.py --no-cpu-baseline --no-recommend --no-graph > "$out/bench_eager.json" 2> "$out/bench_eager.err"; echo "bench eager rc=$?" >> "$out/summary.txt"
tail -n 12 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
tail -n 5 "$out/bench.err" | cut -c1-300 >> "$out/summary.txt"
python - >> "$out/summary.txt" <<'PY'
import json
for f in ("bench","bench_eager"):
    try:
        line=[l for l in open(f"gpurun_out/r02t/{f}.json") if l.startswith('{"metric"')][-1]
        d=json.loads(line)
        print(f, "step", d["ms_per_step"], d["config"]["launch"], {k: v["mean_ms"] for k,v in d["kernels"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
cat "$out/summary.txt"
