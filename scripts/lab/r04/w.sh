#!/bin/bash
# r04w: score_topk lockstep window sweep: time + fabric reads per variant
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04w
mkdir -p "$out"
for v in 256_1 256_2 512_1 128_2 512_2 1024_1; do
  export LIBRECO_HIP_LIB=$PWD/build/lab/libreco_tk_$v.so
  echo "== $v"
  timeout 300 python scripts/score_topk_traffic.py 2>&1 | grep "ms/pass"
  bash scripts/pmc_cmd.sh r04topk_$v "python scripts/score_topk_traffic.py --once" "TCC_EA0_RDREQ_sum" 2>&1 | grep -E "score_topk" | cut -c60-200
done
