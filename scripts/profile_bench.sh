#!/bin/bash
# rocprofv3 kernel-trace of the bench command; summaries are copied into profiles/ by hand.
# usage (on the GPU box, via gpurun): bash scripts/profile_bench.sh <tag> [bench args...]
set -e
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -f csv rocpd -d $OUT -o bench -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_stdout.log 2>&1 || true
cd $OLDPWD
ls -R $OUT | head -30
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
echo "== $F"; head -40 "$F"
