#!/bin/bash
# kernel durations of one command from a rocprofv3 kernel trace (no counters): mean / min per kernel name
# usage: bash scripts/trace_cmd.sh <tag> "<command>" [name filter]
TAG=$1; CMD=$2; FILTER=${3:-lr::}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT
(cd /tmp && timeout ${TRACE_TIMEOUT:-120} rocprofv3 --kernel-trace -f csv -d $OUT -o kt -- bash -c "cd $ROOT && $CMD" > $OUT/stdout.log 2>&1)
python - "$OUT" "$FILTER" <<'PY'
import csv, sys, glob, collections
out, flt = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
if not f: print("no kernel trace in", out); sys.exit()
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f[0]))
               if flt in r["Kernel_Name"]), key=lambda x: x[0])
# launches of one kernel without a pause of more than 1 ms between them = one run (one bench variant)
runs, last = [], {}
for st, en, n in rows:
    d = (en - st) / 1e3
    if n in last and st - last[n][1] < 1_000_000:
        runs[last[n][0]][1].append(d)
        last[n][1] = en
    else:
        runs.append([n, [d]])
        last[n] = [len(runs) - 1, en]
for n, v in runs:
    if len(v) < 3:
        continue
    v2 = v[len(v) // 3:]          # (first launches: cold caches / clocks)
    print(f"{n[:70]:72s} n={len(v):3d} mean {sum(v2)/len(v2):8.1f} us  min {min(v):8.1f}")
PY
