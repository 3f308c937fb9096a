#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / mean / share.
usage: python scripts/rocpd_summary.py <results.db> [top_n] > profiles/<name>.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration),"
                  " max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name"
                  " order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
print(f"| kernel | calls | total ms | mean us | min us | max us | % | vgpr | agpr | lds |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in rows[:top]:
    name = r[0] if len(r[0]) < 110 else r[0][:107] + "..."
    print(f"| `{name}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.1f} | {r[4]/1e3:.1f} | {r[5]/1e3:.1f} | "
          f"{100*r[2]/total:.1f} | {r[6]} | {r[7]} | {r[8]} |")
print(f"\ntotal kernel time {total/1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct kernels")
