#!/bin/bash
# r02al: YouTubeRetrieval (row f4) parity + API tests
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_youtube_retrieval_gpu.py -m gpu -x -q > gpurun_out/r02al_tests.log 2>&1
tail -45 gpurun_out/r02al_tests.log
