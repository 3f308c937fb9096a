#!/bin/bash
# rocprofv3 --kernel-trace --stats of the four bench lines (same commands as the committed bench JSONs, without the CPU leg)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r03trace
mkdir -p "$out"
export TMPDIR=/tmp
for w in deepfm din twotower lightgcn; do
  steps=20; [ $w = lightgcn ] && steps=5; [ $w = twotower ] && steps=5
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$w -o kt -- python $ROOT/bench.py --workload $w --steps $steps --warmup 3 --no-cpu-baseline --no-recommend --no-dense-adam-line --steady-seconds 0 > $out/prof_$w.log 2>&1)
  echo "$w rc=$?"
  f=$(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/kernel_stats_$w.csv"
  grep "^{" $out/prof_$w.log | tail -1 > $out/bench_$w.json
done
ls -la $out
