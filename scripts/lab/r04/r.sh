#!/bin/bash
# r04r: row-sharded step at world 1 without RCCL self-copies: tests of the sharded paths + bench --force-sharded
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r04r
mkdir -p "$out"
timeout 900 python -m pytest tests/test_dist_api_gpu.py tests/test_sharded_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$out/pytest.log"
timeout 300 python bench.py --force-sharded --steps 20 --warmup 5 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line --steady-seconds 0 > "$out/force_sharded.json" 2> "$out/force_sharded.err"; echo "force-sharded rc=$?"; grep -o '"ms_per_step": [0-9.]*' "$out/force_sharded.json" | head -1
