#!/usr/bin/env python3
"""Per-kernel average of rocprofv3 PMC counters (FETCH_SIZE / WRITE_SIZE are in KiB) from a rocpd database."""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name order by 1"))
for name, cname, n, avg, dur in rows:
    short = name.replace("(anonymous namespace)::", "").split("(")[0][-60:]
    if len(sys.argv) > 2 and not any(k in name for k in sys.argv[2:]):
        continue
    print("%-62s %-11s n=%3d avg=%12.1f KiB  (%.1f MB)  avg_dur=%.1f us" % (short, cname, n, avg, avg * 1024 / 1e6, dur / 1e3))
