#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r03j
mkdir -p "$out"
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_din_fused_gpu.py -m gpu -q --timeout 600 -k "block or k16 or vs_numpy" > "$out/t_k16.log" 2>&1
echo "k16 rc=$? $(tail -n 1 "$out/t_k16.log" | cut -c1-160)" >> "$out/summary.txt"
FIT_NF=20 timeout 900 python scripts/fit_bench.py > "$out/fit_42.txt" 2>&1; echo "fit42 rc=$?" >> "$out/summary.txt"
FIT_NF=100 FIT_N=1000000 timeout 1200 python scripts/fit_bench.py > "$out/fit_202.txt" 2>&1; echo "fit202 rc=$?" >> "$out/summary.txt"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recommend --dense-adam-line > "$out/bench_dense_adam.json" 2> "$out/bench_dense_adam.err"; echo "dense adam rc=$?" >> "$out/summary.txt"
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"
bash scripts/pmc_cmd.sh r03din "python bench.py --workload din --steps 3 --warmup 2 --no-cpu-baseline --no-graph --steady-seconds 0" "$TCC" > "$out/pmc_din.log" 2>&1
bash scripts/pmc_cmd.sh r03lg "python bench.py --workload lightgcn --steps 2 --warmup 1 --no-cpu-baseline --steady-seconds 0" "$TCC" > "$out/pmc_lightgcn.log" 2>&1
bash scripts/pmc_cmd.sh r03tt "python bench.py --workload twotower --steps 2 --warmup 1 --no-cpu-baseline --steady-seconds 0" "$TCC" > "$out/pmc_twotower.log" 2>&1
grep -h "lr::" "$out"/pmc_*.log | cut -c1-400 >> "$out/summary.txt"
tail -5 "$out/fit_42.txt" "$out/fit_202.txt" >> "$out/summary.txt"
cat "$out/summary.txt"
