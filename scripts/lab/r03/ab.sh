#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03ab
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_din_fused_gpu.py tests/test_din_tower_models_gpu.py tests/test_graph_fit_gpu.py tests/test_youtube_retrieval_gpu.py tests/test_sharded_gpu.py -m gpu -q > gpurun_out/r03ab/t.log 2>&1; echo rc=$?; tail -3 gpurun_out/r03ab/t.log
