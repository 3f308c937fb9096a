#!/bin/bash
# r06e: plan stream priority of the row-sharded DeepFM step at world size 1
cd "${GRAFT_REPO_ROOT:-.}"
for p in -1 0; do
LIBRECO_PLAN_STREAM_PRIORITY=$p timeout 300 python bench.py --force-sharded --steps 30 --warmup 10 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('priority $p', d['ms_per_step'], (d.get('steady_state') or {}).get('ms_per_step'))"
done
