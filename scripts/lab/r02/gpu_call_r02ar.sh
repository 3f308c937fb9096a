#!/bin/bash
# r02ar: kernel trace of the DIN cfg-3 train step alone (where does the 1.6 ms go?)
set -x
mkdir -p gpurun_out/r02ar
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OLDPWD/gpurun_out/r02ar/prof -o din -- python $OLDPWD/scripts/model_suite.py din > $OLDPWD/gpurun_out/r02ar/stdout.log 2>&1)
tail -3 gpurun_out/r02ar/stdout.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/r02ar/prof/din_kernel_stats.csv')))
tot = sum(int(r['TotalDurationNs']) for r in rows)
steps = 13
print(f"total kernel time per step: {tot/steps/1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)/steps:.0f} launches")
for r in rows[:28]:
    n = r['Name'].replace('void ', '')[:90]
    print(f"{int(r['Calls'])/steps:6.1f} x {float(r['AverageNs'])/1e3:8.1f} us = {int(r['TotalDurationNs'])/steps/1e3:8.1f} us  {n}")
PY
