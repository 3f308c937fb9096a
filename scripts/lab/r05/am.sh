#!/bin/bash
# r05am: attention kernels' persistent grid capped at 2 / 4 / 8 workgroups per CU (DIN cfg 3 step)
mkdir -p gpurun_out/r05am
for v in base g4 g8; do
  if [ $v = base ]; then unset LIBRECO_HIP_LIB; else export LIBRECO_HIP_LIB=$PWD/build/lab/libreco_din_$v.so; fi
  timeout 300 python bench.py --workload din --steps 40 --warmup 10 --no-cpu-baseline --steady-seconds 1 > gpurun_out/r05am/din_$v.json 2> gpurun_out/r05am/din_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05am/din_$v.json").read().strip().splitlines()[-1])
k=d.get("kernels",{})
print("$v", d.get("ms_per_step"), d.get("steady_state"), {n:k[n]["mean_ms"] for n in k if "din_attn" in n})
PY
done
