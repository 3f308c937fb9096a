#!/bin/bash
# PMC counters for one kernbench target: separate passes (no trace domains besides kernel-trace).
# usage: bash scripts/pmc.sh <tag> <kbench target> "<COUNTERS pass1>" ["<COUNTERS pass2>" ...]
TAG=$1; TGT=$2; shift 2
export TMPDIR=/tmp
i=0
for C in "$@"; do
  OUT=$PWD/gpurun_out/pmc_${TAG}_$i
  mkdir -p $OUT
  (cd /tmp && rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT -o pmc -- python $OLDPWD/scripts/kbench.py $TGT 3 > $OUT/stdout.log 2>&1)
  python - "$OUT" <<'PY'
import csv, sys, glob, collections
out=sys.argv[1]
f=glob.glob(out+"/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv in", out); sys.exit()
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "lr::" in k:
        print(k, {c: (len(x), sum(x)/len(x)) for c,x in v.items()})
PY
  i=$((i+1))
done
