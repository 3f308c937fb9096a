#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02b
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_deepfm_fused_gpu.py -q --timeout 300 > "$out/fused_tests.log" 2>&1; echo "fused tests rc=$?" >> "$out/summary.txt"
timeout 300 python scripts/fused_kbench.py all 5 > "$out/kbench_ts32.log" 2>&1; echo "kbench32 rc=$?" >> "$out/summary.txt"
LIBRECO_L1_TILE=64 timeout 300 python scripts/fused_kbench.py all 5 > "$out/kbench_ts64.log" 2>&1; echo "kbench64 rc=$?" >> "$out/summary.txt"
common="--no-cpu-baseline --no-recommend --steps 30 --warmup 10"
timeout 300 python bench.py $common > "$out/bench_graph.json" 2> "$out/bench_graph.err"; echo "bench graph rc=$?" >> "$out/summary.txt"
timeout 300 python bench.py $common --no-graph > "$out/bench_eager.json" 2> "$out/bench_eager.err"; echo "bench eager rc=$?" >> "$out/summary.txt"
timeout 600 python scripts/diag_fullsize.py --oracle > "$out/diag.log" 2>&1; echo "diag rc=$?" >> "$out/summary.txt"
timeout 600 python -m pytest tests/test_fullsize_parity_gpu.py -q --timeout 400 > "$out/parity_tests.log" 2>&1; echo "parity tests rc=$?" >> "$out/summary.txt"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$out/prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-recommend > $OLDPWD/$out/prof_stdout.log 2>&1)
F=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && head -70 "$F" > "$out/kernel_stats_head.csv"
for f in fused_tests kbench_ts32 kbench_ts64 diag parity_tests; do echo "== $f"; tail -n 22 "$out/$f.log"; done >> "$out/summary.txt" 2>/dev/null
cat "$out/bench_graph.json" "$out/bench_eager.json" >> "$out/summary.txt"
tail -n 5 "$out/bench_graph.err" >> "$out/summary.txt"
tail -n 150 "$out/summary.txt"
