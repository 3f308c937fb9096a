#!/bin/bash
# r02au: regression test of graph replays with alternating batch shapes + the fit-level bench that exposed the bug
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_deepfm_fused_gpu.py -m gpu -x -q -k "graph" > gpurun_out/r02au_tests.log 2>&1
tail -15 gpurun_out/r02au_tests.log
timeout 600 python -u scripts/fit_bench.py > gpurun_out/r02au_fit_bench.txt 2>&1
tail -8 gpurun_out/r02au_fit_bench.txt
