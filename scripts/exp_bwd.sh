#!/bin/bash
# layout / cache-policy experiments for the fused FM backward (variant libs under build/exp)
for v in base nt aos aosnt; do
  export LIBRECO_HIP_LIB=$PWD/build/exp/lib_$v.so
  if [[ $v == aos* ]]; then export KB_AOS=1; else unset KB_AOS; fi
  echo "== $v"; python scripts/kbench.py bwd 5 2>&1 | tail -3
done
