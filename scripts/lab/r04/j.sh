#!/bin/bash
# r04j: feature-block DeepFM step + YouTubeRetrieval.rebuild_model tests
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04j
mkdir -p "$out"
timeout 900 python -m pytest tests/test_feat_block_gpu.py tests/test_retrain_gpu.py tests/test_feat_api_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -30 "$out/pytest.log"
