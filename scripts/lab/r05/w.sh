#!/bin/bash
# r05w: durations of the two bitmap products of a LightGCN step (kernel trace)
out=gpurun_out/r05w; mkdir -p $out
export TMPDIR=/tmp
ROOT=$PWD
(cd /tmp && timeout 500 rocprofv3 --kernel-trace -f csv -d $ROOT/$out/trace -o kt -- bash -c "cd $ROOT && python bench.py --workload lightgcn --steps 4 --warmup 2 --no-cpu-baseline --steady-seconds 0 > $ROOT/$out/bench.json 2> $ROOT/$out/bench.err") > $out/trace.log 2>&1
python - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/r05w/trace/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)) if 'spmm_' in r['Kernel_Name'] or 'adam_dense' in r['Kernel_Name']), key=lambda x:x[0])
for st,en,n in rows[-40:]:
    print(f"{(en-st)/1e6:9.3f} ms  {n[:70]}")
PY
find $out/trace -name "*.csv" -delete
