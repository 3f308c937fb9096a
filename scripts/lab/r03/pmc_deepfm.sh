#!/bin/bash
# HBM-side traffic of the default line's kernels on the final tree (train step + recommend leg), one PMC pass
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03pmcdeepfm
bash scripts/pmc_cmd.sh r03deepfm "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense-adam-line --steady-seconds 0 --no-graph" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" > gpurun_out/r03pmcdeepfm/pmc.log 2>&1
cut -c1-400 gpurun_out/r03pmcdeepfm/pmc.log | head -60
