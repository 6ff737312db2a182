#!/usr/bin/env python3
"""per-kernel statistics from a rocprofv3 rocpd SQLite database:  python tools/rocpd_stats.py results.db [csv_out]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                   f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
for name, calls, tot, mn, mx in rows:
    lines.append(f'"{name}",{calls},{tot},{tot / calls:.1f},{100.0 * tot / total:.2f},{mn},{mx}')
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
for name, calls, tot, mn, mx in rows:
    print(f"{name[:100]:100s} calls={calls:5d} avg_us={tot / calls / 1e3:10.2f} min_us={mn / 1e3:9.2f} tot_ms={tot / 1e6:9.3f}")
