#!/bin/bash
# r05ab: nonzeros in flight of the bitmap-filtered product (8 / 16 / 32): kernel trace of the LightGCN step
out=gpurun_out/r05ab; mkdir -p $out
export TMPDIR=/tmp
ROOT=$PWD
for v in default mw8 mw32; do
  lib=$ROOT/build/lab/libreco_$v.so; [ $v = default ] && lib=$ROOT/librecommender_amd/lib/liblibreco_hip.so
  rm -rf $out/trace_$v
  (cd /tmp && LIBRECO_HIP_LIB=$lib timeout 400 rocprofv3 --kernel-trace -f csv -d $ROOT/$out/trace_$v -o kt -- bash -c "cd $ROOT && python bench.py --workload lightgcn --steps 3 --warmup 2 --no-cpu-baseline --steady-seconds 0 > $ROOT/$out/bench_$v.json 2> $ROOT/$out/bench_$v.err") > $out/trace_$v.log 2>&1
  python - $v <<'PY'
import csv, glob, sys, collections
v=sys.argv[1]
f=glob.glob(f'gpurun_out/r05ab/trace_{v}/**/*kernel_trace.csv', recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'spmm_bucketed' in n: agg[n.split('(')[0][-45:]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
for k,x in agg.items():
    x=x[len(x)//2:]
    print(v, k, len(x), 'median', sorted(x)[len(x)//2], 'all', [round(t,1) for t in x[-8:]])
PY
  find $out/trace_$v -name "*.csv" -delete
done
