#!/bin/bash
# r04y: GPU suite from the graph-node tests on (the files before them passed in the previous call)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04y
mkdir -p "$out"
t0=$(date +%s)
files=$(ls tests/test_*gpu*.py | sort | awk '$0 >= "tests/test_graph_nodes_gpu.py"')
timeout 1700 python -m pytest $files -m gpu -q -x > "$out/pytest_gpu2.log" 2>&1; echo "pytest rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -6 "$out/pytest_gpu2.log" | cut -c1-300
