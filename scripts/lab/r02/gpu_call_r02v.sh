#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02v
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_ops_gpu.py tests/test_softmax_ce_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
timeout 300 python bench.py --no-cpu-baseline --no-recommend --force-sharded > "$out/sh.json" 2> "$out/sh.err"; echo "bench rc=$?" >> "$out/summary.txt"
tail -n 4 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
python - >> "$out/summary.txt" <<'PY'
import json
line=[l for l in open("gpurun_out/r02v/sh.json") if l.startswith('{"metric"')][-1]
d=json.loads(line)
print("sharded step", d["ms_per_step"], {k: (v["mean_ms"], v["launches"]) for k,v in d["kernels"].items()})
PY
cat "$out/summary.txt"
