#!/bin/bash
# first GPU call of round 2: new kernels' tests, kernel timings, bench (fused / unfused), profile
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02a
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_deepfm_fused_gpu.py -q -x --timeout 300 > "$out/fused_tests.log" 2>&1; echo "fused tests rc=$?" >> "$out/summary.txt"
timeout 300 python scripts/fused_kbench.py all 5 > "$out/kbench.log" 2>&1; echo "kbench rc=$?" >> "$out/summary.txt"
common="--no-cpu-baseline --no-recommend --steps 20 --warmup 10"
timeout 300 python bench.py $common > "$out/bench_fused.json" 2> "$out/bench_fused.err"; echo "bench fused rc=$?" >> "$out/summary.txt"
timeout 300 python bench.py $common --unfused > "$out/bench_unfused.json" 2> "$out/bench_unfused.err"; echo "bench unfused rc=$?" >> "$out/summary.txt"
timeout 600 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_oov_values_gpu.py -q --timeout 400 > "$out/parity_tests.log" 2>&1; echo "parity tests rc=$?" >> "$out/summary.txt"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 400 --deselect tests/test_deepfm_fused_gpu.py --deselect tests/test_fullsize_parity_gpu.py --deselect tests/test_oov_values_gpu.py > "$out/pytest_gpu.log" 2>&1; echo "pytest -m gpu rc=$?" >> "$out/summary.txt"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$out/prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-recommend > $OLDPWD/$out/prof_stdout.log 2>&1)
F=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -60 "$F" > "$out/kernel_stats_head.csv"
for f in fused_tests kbench parity_tests pytest_gpu; do echo "== $f"; tail -n 25 "$out/$f.log"; done >> "$out/summary.txt" 2>/dev/null
cat "$out/bench_fused.json" "$out/bench_unfused.json" >> "$out/summary.txt"
tail -n 5 "$out/bench_fused.err" >> "$out/summary.txt"
cat "$out/summary.txt" | tail -n 150
