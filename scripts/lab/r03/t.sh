#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=$PWD/gpurun_out/r03t
mkdir -p "$out"
repo=$PWD
cd /tmp && export TMPDIR=/tmp
for tag in eager hipGraph; do
  FIT_BENCH_ONLY="device loader, $tag" timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_$tag -o kt -- python -u -W ignore $repo/scripts/din_fit_bench.py > "$out/fit_$tag.txt" 2>&1
  echo "$tag rc=$?"
  grep -E "epoch|interactions" "$out/fit_$tag.txt" | cut -c1-200
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  cp "$f" "$out/kernel_stats_$tag.csv"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot / 1e6, 1), "dispatches", sum(int(r["Calls"]) for r in rows))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):7d} {float(r["TotalDurationNs"]) / 1e6:9.2f} ms {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
done
