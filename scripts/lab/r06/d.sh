#!/bin/bash
# r06d: the row-sharded DeepFM step at world size 1 (bench.py --force-sharded): un-profiled time, then a kernel timeline of one step
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD; out=$ROOT/gpurun_out/r06d; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python bench.py --force-sharded --steps 30 --warmup 10 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line > $out/fs.json 2> $out/fs.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06d/fs.json").read().strip().splitlines()[-1])
print("force-sharded ms_per_step", d.get("ms_per_step"), "steady", (d.get("steady_state") or {}).get("ms_per_step"), d.get("config",{}).get("launch"))
for k,v in d.get("kernels",{}).items(): print("   ",k,v.get("launches"),v.get("mean_ms"))
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -f csv -d $out/trace -o fs -- bash -c "cd $ROOT && python bench.py --force-sharded --steps 12 --warmup 4 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line --steady-seconds 0" > $out/trace.log 2>&1)
python scripts/trace_timeline.py $(find $out/trace -name "*kernel_trace.csv" | head -1) idx_transpose 8 > $out/timeline.txt 2>&1
head -90 $out/timeline.txt | cut -c1-200
