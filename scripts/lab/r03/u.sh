#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03u
timeout 600 python -u -W ignore scripts/lab/r03/u.py > gpurun_out/r03u/prof.txt 2>&1; echo rc=$?
grep -v Warning gpurun_out/r03u/prof.txt | head -150
