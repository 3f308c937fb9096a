#!/bin/bash
# r05a: first run of the split-bf16 first-layer kernels (parity + timing) and the 100 M-item scoring parity test
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r05a; mkdir -p $out
t0=$(date +%s)
timeout 600 python -m pytest tests/test_l1_split_bf16_gpu.py -x -q -m gpu > $out/sb_tests.log 2>&1; echo "sb tests rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -25 $out/sb_tests.log | cut -c1-300
t0=$(date +%s)
timeout 400 python scripts/l1_sb_kbench.py > $out/kbench.log 2>&1; echo "kbench rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -30 $out/kbench.log | cut -c1-260
t0=$(date +%s)
timeout 600 python -m pytest tests/test_fullsize_parity_gpu.py::test_score_topk_100m_vs_fp64 tests/test_score_topk_gpu.py::test_lockstep_give_up_path_changes_nothing -x -q -m gpu > $out/topk_tests.log 2>&1; echo "topk tests rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -15 $out/topk_tests.log | cut -c1-300
