#!/bin/bash
# r02ao: embedding server round trip
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_embed_server_gpu.py -m gpu -x -q > gpurun_out/r02ao_tests.log 2>&1
tail -45 gpurun_out/r02ao_tests.log
