#!/bin/bash
# r05ap: row-partitioned LightGCN at world size 1, one-piece (chunks = 1) against column-blocked in 4, same box
mkdir -p gpurun_out/r05ap
for c in 1 4; do
LIBRECO_LGCN_CHUNKS=$c timeout 300 python bench.py --workload lightgcn --force-sharded --steps 5 --warmup 2 --no-cpu-baseline --steady-seconds 0 > gpurun_out/r05ap/lgcn_c$c.json 2> gpurun_out/r05ap/lgcn_c$c.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r05ap/lgcn_c$c.json").read().strip().splitlines()[-1])
print("lightgcn sharded W=1 chunks=$c", d.get("ms_per_step"), d.get("error"))
PY
done
