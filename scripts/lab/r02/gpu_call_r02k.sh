#!/bin/bash
# fold kernels: tests + bench; --force-sharded glue time; DIN model step; TCC counters of DIN / softmax-CE / SpMM
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02k
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python __graft_entry__.py smoke > "$out/smoke.log" 2>&1; echo "smoke rc=$?" >> "$out/summary.txt"
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_fullsize_parity_gpu.py -m gpu -q -x --timeout 600 > "$out/tests.log" 2>&1; echo "fused tests rc=$?" >> "$out/summary.txt"
timeout 300 python bench.py --no-cpu-baseline --no-recommend > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/summary.txt"
timeout 300 python bench.py --no-cpu-baseline --no-recommend --force-sharded > "$out/bench_sharded.json" 2> "$out/bench_sharded.err"; echo "bench force-sharded rc=$?" >> "$out/summary.txt"
timeout 300 python scripts/model_suite.py din > "$out/model_din.log" 2>&1
TCC="TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum"
bash scripts/pmc_cmd.sh din2 "python scripts/kern_suite.py din" "$TCC" > "$out/pmc_din.log" 2>&1
bash scripts/pmc_cmd.sh sce2 "python scripts/sce_bench.py 32768 128 2" "$TCC" > "$out/pmc_sce.log" 2>&1
bash scripts/pmc_cmd.sh spmm2 "python scripts/kern_suite.py spmm" "$TCC" > "$out/pmc_spmm.log" 2>&1
tail -n 3 "$out/smoke.log" | cut -c1-300 >> "$out/summary.txt"
tail -n 6 "$out/tests.log" | cut -c1-300 >> "$out/summary.txt"
cut -c1-3500 "$out/bench.json" >> "$out/summary.txt"; tail -n 2 "$out/bench.err" >> "$out/summary.txt"
cut -c1-3500 "$out/bench_sharded.json" >> "$out/summary.txt"; tail -n 4 "$out/bench_sharded.err" >> "$out/summary.txt"
grep -h "^din\|^twotower" "$out/model_din.log" >> "$out/summary.txt"
grep -h "lr::" "$out/pmc_din.log" "$out/pmc_sce.log" "$out/pmc_spmm.log" | cut -c1-400 >> "$out/summary.txt"
cat "$out/summary.txt"
