#!/bin/bash
# r06b: DIN step with / without the forked segment build (does the radix sort overlap the main branch in the replayed graph?)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r06b
for f in 1 0; do
  LIBRECO_DIN_FORK=$f timeout 300 python bench.py --workload din --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r06b/din_fork$f.json 2> gpurun_out/r06b/din_fork$f.err
  python - $f <<'PY'
import json,sys
f=sys.argv[1]
d=json.loads(open(f"gpurun_out/r06b/din_fork{f}.json").read().strip().splitlines()[-1])
print("fork",f,"ms_per_step",d.get("ms_per_step"),"steady",d.get("steady_state",{}).get("ms_per_step"), "sum_kernel_ms", d.get("sum_kernel_ms"))
for k,v in d.get("kernels",{}).items(): print("   ",k,v.get("launches"),v.get("mean_ms"))
PY
done
