#!/usr/bin/env python
"""One step of a rocprofv3 --kernel-trace CSV as a timeline: start (us from the step's first kernel), duration, gap to the
previous kernel on the same hardware queue, idle time of the whole GPU before it, queue / stream, kernel name.
usage: trace_timeline.py <kernel_trace.csv[.gz]> <substring of the kernel that starts a step> [min_us]"""
import csv
import gzip
import io
import re
import sys

path, key = sys.argv[1], sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
f = io.TextIOWrapper(gzip.open(path)) if path.endswith(".gz") else open(path)
rows = list(csv.DictReader(f))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
a, b = marks[-4], marks[-3]
T0 = rows[a]["s"]
print(f"step span {(rows[b]['s'] - T0) / 1000:.1f} us ({b - a} kernels)")
prev_end, busy, cur_e, small, small_n = {}, 0, T0, 0.0, 0
for r in rows[a:b]:
    nm = re.sub(r"void |at::native::|\(anonymous namespace\)::|rocprim::ROCPRIM_\d+_NS::detail::", "", r["Kernel_Name"])[:72]
    q = r["Queue_Id"]
    gap = (r["s"] - prev_end.get(q, r["s"])) / 1000
    idle = max(0, r["s"] - cur_e) / 1000
    dur = (r["e"] - r["s"]) / 1000
    if dur >= min_us or idle >= 10:
        if small_n:
            print(f"{'':>9} {small:8.1f}  ... {small_n} kernels under {min_us:g} us")
            small, small_n = 0.0, 0
        print(f"{(r['s'] - T0) / 1000:9.1f} {dur:8.1f} qgap{gap:7.1f} idle{idle:6.1f} q{q} s{r['Stream_Id']} {nm}")
    else:
        small += dur
        small_n += 1
    prev_end[q] = max(prev_end.get(q, 0), r["e"])
    busy += max(0, r["e"] - max(cur_e, r["s"]))
    cur_e = max(cur_e, r["e"])
if small_n:
    print(f"{'':>9} {small:8.1f}  ... {small_n} kernels under {min_us:g} us")
print(f"gpu busy (union) {busy / 1000:.1f} us")
