#!/usr/bin/env python3
"""Step durations (end of one adam_kernel to the end of the next) over a whole rocprofv3 run, and for the slowest step its largest
main-stream gaps: prof_steps.py results.db"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, stream_id, start, end from kernels order by start"))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
d = [(rows[adam[k + 1]][3] - rows[adam[k]][3]) / 1e6 for k in range(len(adam) - 1)]
print("steps:", " ".join("%.2f" % v for v in d))
k = max(range(len(d)), key=lambda i: d[i] if i > 15 else 0)
print("slowest late step %d: %.2f ms" % (k, d[k]))
seg = [r for r in rows[adam[k] + 1: adam[k + 1] + 1] if r[1] == rows[adam[k]][1]]
prev = rows[adam[k]][3]
gaps = []
for r in seg:
    gaps.append(((r[2] - prev) / 1e3, r[0].split("(")[0][-50:], (r[3] - r[2]) / 1e3))
    prev = r[3]
for g in sorted(gaps, reverse=True)[:12]:
    print("  gap %8.1f us before %-50s (dur %.1f us)" % g)
print("  busy %.1f us in %d kernels" % (sum(g[2] for g in gaps), len(gaps)))
