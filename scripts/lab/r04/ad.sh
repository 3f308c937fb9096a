#!/bin/bash
# r04ad: experimental split-bf16 first-layer forward: parity + timing at cfg 2's shape
set -u
cd "${GRAFT_REPO_ROOT:-.}"
timeout 300 python -m pytest tests/test_l1_split_bf16_gpu.py -x -q -m gpu 2>&1 | tail -15 | cut -c1-250
timeout 300 python scripts/l1_sb_kbench.py 2>&1 | tail -10
