#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04ac
mkdir -p "$out"
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_dist_api_gpu.py tests/test_owner_partition_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -12 "$out/pytest.log" | cut -c1-250
timeout 300 python bench.py --force-sharded --steps 20 --warmup 5 --no-cpu-baseline --no-recommend --no-dense-adam-line --steady-seconds 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
