#!/bin/bash
# r02am: Transformer (row f4) parity + API tests
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_din_tower_models_gpu.py tests/test_feat_api_gpu.py -m gpu -x -q > gpurun_out/r02am_tests.log 2>&1
tail -45 gpurun_out/r02am_tests.log
