#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03h
timeout 300 python scripts/din_kbench.py > gpurun_out/r03h/din_kbench.txt 2>&1
cat gpurun_out/r03h/din_kbench.txt
