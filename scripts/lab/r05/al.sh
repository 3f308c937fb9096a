#!/bin/bash
mkdir -p gpurun_out/r05al
timeout 300 python scripts/lab/r05/din_tail_marks.py 2>&1 | grep -v -i "rccl\|nccl\|warn" | tail -24 > gpurun_out/r05al/marks.txt
cat gpurun_out/r05al/marks.txt
