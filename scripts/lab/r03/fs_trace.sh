#!/bin/bash
# kernel trace of the row-sharded DeepFM step at world size 1 (RCCL self-exchange): where the 6.1 ms go
set -u
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$PWD
out=$ROOT/gpurun_out/r03fstrace
mkdir -p "$out"
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_fs -o kt -- python $ROOT/bench.py --force-sharded --steps 10 --warmup 3 --no-cpu-baseline --no-recommend --steady-seconds 0 > $out/prof.log 2>&1); echo "prof rc=$?"
f=$(find /tmp/prof_fs -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/kernel_stats.csv"
f=$(find /tmp/prof_fs -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python - "$f" "$out/trace_tail.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last ~2 steps' worth of launches, with gaps
tail = rows[-400:]
with open(sys.argv[2], "w") as f:
    prev = None
    for r in tail:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0 if prev is None else s - prev
        f.write(f'{r["Kernel_Name"][:90]},{(e - s) / 1e3:.1f},{gap / 1e3:.1f},{r.get("Stream_Id", "")}\n')
        prev = max(e, prev or 0)
PY
grep "^{" $out/prof.log | tail -1 | grep -o '"ms_per_step": [0-9.]*'
