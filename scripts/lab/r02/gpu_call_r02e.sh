#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02e
mkdir -p "$out"
timeout 900 python -m pytest tests/test_deepfm_fused_gpu.py -q --timeout 300 > "$out/fused_tests.log" 2>&1; echo "fused tests rc=$?" >> "$out/summary.txt"
timeout 600 python -m pytest tests/test_fullsize_parity_gpu.py -q --timeout 400 -k deepfm > "$out/parity_tests.log" 2>&1; echo "parity tests rc=$?" >> "$out/summary.txt"
for A in 0; do
  echo "== ablate $A" >> "$out/prio.log"
  LIBRECO_L1_ABLATE=$A timeout 200 python scripts/fused_kbench.py l1 5 2>&1 | grep -E "ms:" >> "$out/prio.log"
done
common="--no-cpu-baseline --no-recommend --steps 30 --warmup 10"
timeout 300 python bench.py $common > "$out/bench_graph.json" 2> "$out/bench_graph.err"; echo "bench graph rc=$?" >> "$out/summary.txt"
timeout 300 python bench.py $common --no-graph > "$out/bench_eager.json" 2> "$out/bench_eager.err"; echo "bench eager rc=$?" >> "$out/summary.txt"
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/$out/prof -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-recommend > $OLDPWD/$out/prof_stdout.log 2>&1)
for f in fused_tests parity_tests prio; do echo "== $f"; tail -n 30 "$out/$f.log" | cut -c1-400; done >> "$out/summary.txt" 2>/dev/null
cat "$out/bench_graph.json" "$out/bench_eager.json" | cut -c1-3000 >> "$out/summary.txt"
tail -n 5 "$out/bench_graph.err" >> "$out/summary.txt"
tail -n 150 "$out/summary.txt"
