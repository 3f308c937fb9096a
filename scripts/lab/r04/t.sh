#!/bin/bash
# r04t: kernel timeline of one workload step (din by default)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
W=${1:-din}
out=$ROOT/gpurun_out/r04t
mkdir -p "$out"
export TMPDIR=/tmp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_r04 -o kt -- python $ROOT/bench.py --workload $W --steps 12 --warmup 3 --no-cpu-baseline > $out/prof_$W.log 2>&1)
tail -2 $out/prof_$W.log | cut -c1-600
f=$(find /tmp/prof_r04 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && gzip -c "$f" > "$out/kernel_trace_$W.csv.gz"
f=$(find /tmp/prof_r04 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats_$W.csv"
