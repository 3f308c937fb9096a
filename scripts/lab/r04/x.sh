#!/bin/bash
# r04x: own radix sort in lr_segments_build: parity suites that build segments + DIN step time
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04x
mkdir -p "$out"
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_din_fused_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$out/pytest.log" | cut -c1-250
timeout 300 python bench.py --workload din --no-cpu-baseline 2> "$out/din.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('din ms', d['ms_per_step'], 'steady', d.get('steady_state'))"
bash scripts/lab/r04/t.sh din > /dev/null 2>&1
