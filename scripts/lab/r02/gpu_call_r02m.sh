#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02m
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_sharded_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
LIBRECO_ROWS_EARLY=0 timeout 200 python scripts/fused_kbench.py adam 8 > "$out/adam_late.log" 2>&1
LIBRECO_ROWS_EARLY=1 timeout 200 python scripts/fused_kbench.py adam 8 > "$out/adam_early.log" 2>&1
LIBRECO_ROWS_EARLY=0 timeout 300 python bench.py --no-cpu-baseline --no-recommend --force-sharded > "$out/sh_late.json" 2> "$out/sh_late.err"
LIBRECO_ROWS_EARLY=1 timeout 300 python bench.py --no-cpu-baseline --no-recommend --force-sharded > "$out/sh_early.json" 2> "$out/sh_early.err"
tail -n 4 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
grep -h "^adam" "$out/adam_late.log" "$out/adam_early.log" >> "$out/summary.txt"
python - >> "$out/summary.txt" <<'PY'
import json
for f in ("sh_late","sh_early"):
    try:
        line=[l for l in open(f"gpurun_out/r02m/{f}.json") if l.startswith('{"metric"')][-1]
        d=json.loads(line)
        print(f, d["ms_per_step"], {k: v["mean_ms"] for k,v in d["kernels"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
cat "$out/summary.txt"
