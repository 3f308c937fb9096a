#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
timeout 600 python -m pytest tests/test_l1_split_bf16_gpu.py tests/test_deepfm_fused_gpu.py tests/test_l1_wide_gpu.py tests/test_owner_partition_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
