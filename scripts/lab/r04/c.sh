#!/bin/bash
# r04c: wide dgrad parity + timing; ablations of the wide forward kernel (what is its time made of?)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04c
mkdir -p "$out"
timeout 600 python -m pytest tests/test_l1_wide_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$out/pytest.log"
for t in 32 64; do
  timeout 200 python scripts/fused_kbench.py dgrad 10 --tile $t > "$out/kbench_dgrad_$t.txt" 2>&1; tail -1 "$out/kbench_dgrad_$t.txt"
done
for n in 1 2 3 4 8 16; do
  LIBRECO_HIP_LIB=$PWD/build/lab/libreco_ab$n.so timeout 200 python scripts/fused_kbench.py fwd 10 --tile 64 > "$out/ab_$n.txt" 2>&1; echo "ablate $n: $(tail -1 $out/ab_$n.txt)"
done
