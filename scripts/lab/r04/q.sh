#!/bin/bash
# r04q: where the row-sharded step's time goes at world 1: kernel trace (timeline) of bench.py --force-sharded
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r04q
mkdir -p "$out"
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_r04 -o kt -- python $ROOT/bench.py --force-sharded --steps 12 --warmup 3 --no-cpu-baseline --no-recommend --no-workloads --no-dense-adam-line --steady-seconds 0 > $out/prof.log 2>&1)
f=$(find /tmp/prof_r04 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats_sharded.csv"
f=$(find /tmp/prof_r04 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && gzip -c "$f" > "$out/kernel_trace_sharded.csv.gz"; ls -la $out
