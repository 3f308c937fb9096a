#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03ac
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_lightgcn_gpu.py tests/test_zz_ngcf_gpu.py -m gpu -q > gpurun_out/r03ac/t.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r03ac/t.log
bash scripts/pmc_cmd.sh r03lg2 "python bench.py --workload lightgcn --steps 2 --warmup 1 --no-cpu-baseline --steady-seconds 0" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" > gpurun_out/r03ac/pmc_lightgcn.log 2>&1
grep -E "spmm|adam_dense" gpurun_out/r03ac/pmc_lightgcn.log | cut -c1-320
timeout 1200 python bench.py --workload lightgcn --steps 20 --warmup 5 > gpurun_out/r03ac/bench_lightgcn.json 2> gpurun_out/r03ac/bench_lightgcn.err; echo "bench rc=$?"
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03ac/bench_lightgcn.json | head -1
