#!/bin/bash
# r02ah: device-side catalogue rows (DIN full-catalogue ranking) + cache invalidation tests; DIN recommend timing
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_feat_api_gpu.py tests/test_api_gpu.py tests/test_retrain_gpu.py -m gpu -x -q > gpurun_out/r02ah_tests.log 2>&1
tail -5 gpurun_out/r02ah_tests.log
timeout 600 python scripts/din_recommend_bench.py > gpurun_out/r02ah_din_rec.txt 2>&1
tail -8 gpurun_out/r02ah_din_rec.txt
