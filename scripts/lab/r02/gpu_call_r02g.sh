#!/bin/bash
# pipelined rows_adam: tests + A/B against the plain kernel + bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02g
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_fullsize_parity_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
timeout 200 python scripts/fused_kbench.py adam 8 > "$out/adam_pipe.log" 2>&1
LIBRECO_ROWS_PLAIN=1 timeout 200 python scripts/fused_kbench.py adam 8 > "$out/adam_plain.log" 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-recommend > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/summary.txt"
tail -n 8 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
grep -h "adam" "$out/adam_pipe.log" "$out/adam_plain.log" >> "$out/summary.txt"
cut -c1-3000 "$out/bench.json" >> "$out/summary.txt"
tail -n 3 "$out/bench.err" >> "$out/summary.txt"
cat "$out/summary.txt"
