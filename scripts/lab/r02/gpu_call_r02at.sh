#!/bin/bash
# r02at: DeepFM.fit epoch throughput through the product API (host / device loader x eager / hipGraph)
set -x
mkdir -p gpurun_out
timeout 900 python scripts/fit_bench.py > gpurun_out/r02at_fit_bench.txt 2>&1
tail -8 gpurun_out/r02at_fit_bench.txt
