#!/bin/bash
# r04d: finer ablation matrix of the wide forward kernel + cycle counters (clock) for the key variants
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04d
mkdir -p "$out"
for n in 15 31 7 11 13 14 3; do
  LIBRECO_HIP_LIB=$PWD/build/lab/libreco_ab$n.so timeout 200 python scripts/fused_kbench.py fwd 10 --tile 64 > "$out/ab_$n.txt" 2>&1; echo "ablate $n: $(tail -1 $out/ab_$n.txt | cut -c1-60)"
done
timeout 200 python scripts/fused_kbench.py fwd 10 --tile 64 > "$out/ab_0.txt" 2>&1; echo "ablate 0: $(tail -1 $out/ab_0.txt | cut -c1-60)"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"
for n in 3 15; do
  LIBRECO_HIP_LIB=$PWD/build/lab/libreco_ab$n.so bash scripts/pmc_cmd.sh r04d_ab$n "python scripts/fused_kbench.py fwd 3 --tile 64" "$P1" > "$out/pmc_ab$n.log" 2>&1
  grep fwd64 "$out/pmc_ab$n.log" | cut -c1-700
done
bash scripts/pmc_cmd.sh r04d_ab0 "python scripts/fused_kbench.py fwd 3 --tile 64" "$P1" > "$out/pmc_ab0.log" 2>&1
grep fwd64 "$out/pmc_ab0.log" | cut -c1-700
