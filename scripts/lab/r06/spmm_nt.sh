#!/bin/bash
# r06: cache-policy variants of the bucketed SpMM on cfg 5's graph: time, then fabric bytes (FETCH_SIZE) of the promising ones
cd "${GRAFT_REPO_ROOT:-.}"
export PYTHONPATH=$PWD
for t in "" s1 c16k s1c16k s1c4k s1c64k; do
  if [ -z "$t" ]; then env -u LIBRECO_HIP_LIB timeout 300 python scripts/lab/r06/spmm_time.py 2>/dev/null | tail -1
  else LIBRECO_HIP_LIB=$PWD/build/lab/libreco_sp_$t.so timeout 300 python scripts/lab/r06/spmm_time.py 2>/dev/null | tail -1; fi
done
for t in "" s1c16k; do
  if [ -z "$t" ]; then unset LIBRECO_HIP_LIB; else export LIBRECO_HIP_LIB=$PWD/build/lab/libreco_sp_$t.so; fi
  SPMM_REPS=2 bash scripts/pmc_cmd.sh r06spmm_$t "python scripts/lab/r06/spmm_time.py" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" 2>&1 | grep -E "spmm_bucketed" | cut -c1-300
done
