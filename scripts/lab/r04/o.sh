#!/bin/bash
# r04o: dropout in the tail kernels: parity vs the oracle with the restated mask + the suites that use the tail
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04o
mkdir -p "$out"
timeout 900 python -m pytest tests/test_tail_dropout_gpu.py tests/test_deepfm_fused_gpu.py tests/test_din_fused_gpu.py tests/test_feat_block_gpu.py tests/test_feat_api_gpu.py tests/test_fm_models_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -25 "$out/pytest.log"
