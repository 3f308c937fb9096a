#!/bin/bash
# r05ao: row-partitioned LightGCN at world size 1 with the slice column-blocked in 4 (the compute side of the chunked form)
mkdir -p gpurun_out/r05ao
LIBRECO_LGCN_CHUNKS=4 timeout 600 python bench.py --workload lightgcn --force-sharded --steps 5 --warmup 2 --no-cpu-baseline --steady-seconds 0 > gpurun_out/r05ao/lgcn_c4.json 2> gpurun_out/r05ao/lgcn_c4.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05ao/lgcn_c4.json").read().strip().splitlines()[-1])
print("lightgcn sharded W=1 chunks=4", d.get("ms_per_step"), d.get("error"))
PY
tail -3 gpurun_out/r05ao/lgcn_c4.err
