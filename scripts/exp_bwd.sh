#!/bin/bash
# experiment: variant libs under build/exp, fused FM backward timing
for v in "$@"; do
  export LIBRECO_HIP_LIB=$PWD/build/exp/lib_$v.so
  echo "== $v"; python scripts/kbench.py bwd 5 2>&1 | tail -1
done
unset LIBRECO_HIP_LIB; echo "== current"; python scripts/kbench.py bwd 5 2>&1 | tail -1
