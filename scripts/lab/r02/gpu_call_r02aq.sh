#!/bin/bash
# r02aq: ShardedDINNet on the HIP kernels (two gloo ranks on one GPU)
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sharded_gpu.py -k din -m gpu -x -q > gpurun_out/r02aq_tests.log 2>&1
tail -45 gpurun_out/r02aq_tests.log
