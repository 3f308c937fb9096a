#!/bin/bash
# r04u: sharded feature layer with the HIP kernels + the bench command line with two ranks on one GPU
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04u
mkdir -p "$out"
timeout 1500 python -m pytest tests/test_dist_api_gpu.py tests/test_bench_cli_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -30 "$out/pytest.log" | cut -c1-300
