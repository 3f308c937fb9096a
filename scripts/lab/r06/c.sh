#!/bin/bash
# r06c: hipGraph runtime knobs of libamdhip64 (strings: DEBUG_CLR_GRAPH_PACKET_CAPTURE, DEBUG_HIP_FORCE_GRAPH_QUEUES,
# DEBUG_HIP_GRAPH_BATCH_SIZE) against the replayed DIN / DeepFM steps
cd "${GRAFT_REPO_ROOT:-.}"
run() {
  for w in din deepfm; do
    env "$@" timeout 300 python bench.py --workload $w --steps 50 --warmup 10 --no-cpu-baseline --no-workloads --no-recommend --no-dense-adam-line 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['config']['workload'][:12], d['ms_per_step'], (d.get('steady_state') or {}).get('ms_per_step'))
except Exception as e: print('   failed', e)"
  done
}
echo "default"; run X=1
echo "PACKET_CAPTURE=1"; run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
echo "PACKET_CAPTURE=0"; run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
echo "FORCE_GRAPH_QUEUES=1"; run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
echo "FORCE_GRAPH_QUEUES=4"; run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
echo "BATCH_SIZE=1"; run DEBUG_HIP_GRAPH_BATCH_SIZE=1
echo "BATCH_SIZE=64"; run DEBUG_HIP_GRAPH_BATCH_SIZE=64
