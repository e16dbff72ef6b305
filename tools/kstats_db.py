#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 results database (rocpd sqlite, what `rocprofv3 --kernel-trace` writes on this
image when no csv output is asked for):  python tools/kstats_db.py <results.db> [passes] [top]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s "
                        f"on d.kernel_id=s.id group by s.kernel_name order by 4 desc"))
tot = sum(r[3] for r in rows)
print(f"{sys.argv[1]}: {tot / 1e6 / passes:.3f} ms of kernel time per pass ({passes} passes), {sum(r[1] for r in rows) / passes:.0f} launches")
for r in rows[:top]:
    print(f"  {r[0][:96]:96s} n={r[1] / passes:6.1f} avg={r[2] / 1e3:8.1f}us tot={r[3] / 1e6 / passes:7.3f}ms {100 * r[3] / tot:5.1f}%")
