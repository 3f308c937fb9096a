#!/bin/bash
# r04k: two-rank checkpoint loaded by one process (+ the other dist API tests on the GPU)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04k
mkdir -p "$out"
timeout 900 python -m pytest tests/test_dist_api_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -30 "$out/pytest.log"
