#!/usr/bin/env python3
"""Per-kernel averages of every counter in rocprofv3 --pmc rocpd databases (development aid): pmc_dump.py a.db [b.db ...] [substring]"""
import sqlite3, sys
dbs = [a for a in sys.argv[1:] if a.endswith(".db")]
sub = next((a for a in sys.argv[1:] if not a.endswith(".db")), "")
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    for name, ctr, n, avg, dur in cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
        if sub in name:
            print("%-70s %-12s n=%3d avg=%14.1f  avg_dur=%9.1f us" % (name.replace("(anonymous namespace)::", "")[:70], ctr, n, avg, dur / 1e3))
