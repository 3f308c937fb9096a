#!/bin/bash
# r04f: issue-cost microbenchmark (what an instruction between two f32 MFMAs costs a wave that is alone on its SIMD)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04f
mkdir -p "$out"
timeout 120 ./build/lab/mfma_issue_probe > "$out/mfma_issue_probe.txt" 2>&1; echo rc=$?
cat "$out/mfma_issue_probe.txt" | cut -c1-230
