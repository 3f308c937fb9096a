#!/bin/bash
mkdir -p gpurun_out/r05ak
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py tests/test_din_fused_gpu.py tests/test_fm_models_gpu.py tests/test_din_gpu.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -4 > gpurun_out/r05ak/tests_fold.log
cat gpurun_out/r05ak/tests_fold.log
