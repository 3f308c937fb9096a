#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02q
mkdir -p "$out"
timeout 900 python -m pytest tests/test_score_topk_gpu.py tests/test_fullsize_parity_gpu.py tests/test_zz_knn_gpu.py -m gpu -q -x --timeout 600 -k "topk or knn or score" > "$out/tests.log" 2>&1; echo "tests rc=$?" >> "$out/summary.txt"
timeout 300 python scripts/kbench.py scoref 6 > "$out/score_pre.log" 2>&1
LIBRECO_TOPK_PREPASS=0 timeout 300 python scripts/kbench.py scoref 6 > "$out/score_nopre.log" 2>&1
tail -n 4 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
echo "with pre-pass:" >> "$out/summary.txt"; grep "score ms" "$out/score_pre.log" | tr '\n' ' ' >> "$out/summary.txt"; echo >> "$out/summary.txt"
echo "without:" >> "$out/summary.txt"; grep "score ms" "$out/score_nopre.log" | tr '\n' ' ' >> "$out/summary.txt"; echo >> "$out/summary.txt"
cat "$out/summary.txt"
