#!/bin/bash
# r02aj: DIN general path (item side features) on the dense-form MFMA attention kernels
set -x
mkdir -p gpurun_out
true
timeout 600 python scripts/din_recommend_bench.py > gpurun_out/r02aj_din_rec.txt 2>&1
tail -60 gpurun_out/r02aj_din_rec.txt
