#!/bin/bash
# PMC counters for one command: separate passes (kernel-trace + pmc only, as the pool requires).
# usage: bash scripts/pmc_cmd.sh <tag> "<command>" "<COUNTERS pass1>" ["<COUNTERS pass2>" ...]
TAG=$1; CMD=$2; shift 2
export TMPDIR=/tmp
ROOT=$PWD
i=0
for C in "$@"; do
  OUT=$ROOT/gpurun_out/pmc_${TAG}_$i
  mkdir -p $OUT
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT -o pmc -- bash -c "cd $ROOT && $CMD" > $OUT/stdout.log 2>&1)
  python - "$OUT" <<'PY'
import csv, sys, glob, collections
out=sys.argv[1]
f=glob.glob(out+"/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv in", out); sys.exit()
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "lr::" in k:
        print(k, {c: (len(x), round(sum(x)/len(x),1)) for c,x in v.items()})
PY
  i=$((i+1))
done
