#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r02ac
mkdir -p "$out"
export TMPDIR=/tmp
ROOT=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$out/prof_din -o din -- python $ROOT/scripts/model_suite.py din > $ROOT/$out/din.log 2>&1)
grep "^din" "$out/din.log" > "$out/summary.txt"
python - >> "$out/summary.txt" <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r02ac/prof_din/din_kernel_stats.csv")))
for r in rows[:32]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['AverageNs'])/1e3:9.2f} {r['Percentage']}")
PY
cat "$out/summary.txt"
