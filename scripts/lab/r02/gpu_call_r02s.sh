#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02s
mkdir -p "$out"
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_sharded_gpu.py tests/test_fullsize_parity_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
timeout 200 python scripts/fused_kbench.py seg 8 2>&1 | grep "^seg" >> "$out/summary.txt"
timeout 200 python scripts/fused_kbench.py stats 8 2>&1 | grep "^stats" >> "$out/summary.txt"
timeout 300 python bench.py --no-cpu-baseline --no-recommend > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/summary.txt"
tail -n 4 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
python - >> "$out/summary.txt" <<'PY'
import json
line=[l for l in open("gpurun_out/r02s/bench.json") if l.startswith('{"metric"')][-1]
d=json.loads(line)
print("step", d["ms_per_step"], {k: v["mean_ms"] for k,v in d["kernels"].items()})
PY
cat "$out/summary.txt"
