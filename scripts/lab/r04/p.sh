#!/bin/bash
# r04p: tail kernels rewritten for wide access (colstats, first_bwd, head): kernel trace of the DeepFM step
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r04p
mkdir -p "$out"
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_r04 -o kt -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line --steady-seconds 0 > $out/prof_deepfm.log 2>&1)
f=$(find /tmp/prof_r04 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats_deepfm.csv" && head -45 "$out/kernel_stats_deepfm.csv" | cut -c1-150
