#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r02d
timeout 600 python scripts/diag_paths2.py 2>&1 | grep -v Warn | tee gpurun_out/r02d/diag_paths2.log | cut -c1-250
