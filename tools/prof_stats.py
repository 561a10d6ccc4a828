#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace): per-kernel count / total / avg / min / max and share."""
import sqlite3
import sys

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3,"
                        " max(vgpr_count), max(lds_size), max(grid_x*1.0/workgroup_x*grid_y/workgroup_y*grid_z/workgroup_z) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
s, e = list(cur.execute("select min(start),max(end) from kernels"))[0]
print("# %s: %d kernel names, busy %.1f ms over a %.1f ms span" % (db.split('/')[-1], len(rows), tot / 1e3, (e - s) / 1e6))
print("%-72s %7s %11s %9s %8s %9s %6s %5s %7s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "share", "vgpr", "lds", "wgs"))
for r in rows[:top]:
    print("%-72s %7d %11.1f %9.2f %8.2f %9.2f %5.1f%% %5d %7d %6d" % (r[0].replace("(anonymous namespace)::", "")[:72], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6] or 0, r[7] or 0, r[8] or 0))
