#!/usr/bin/env python3
"""Timeline of the last training step in a rocprofv3 rocpd database: per stream, every kernel with its start offset,
duration and the idle gap before it; plus per-stream busy / gap totals.  usage: prof_timeline.py results.db [min_gap_us] [all]
(`all`: list the side streams' kernels as well, offsets from the same origin; environment PROF_STEP=<n>: the step that ends with the n-th
optimiser launch of the run instead of the last one -- e.g. a step of bench.py's timed loop rather than of its epoch loop)"""
import os
import sqlite3
import sys

db = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
show_all = "all" in sys.argv[3:]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, stream_id, start, end from kernels order by start"))
adam = [i for i, r in enumerate(rows) if r[0].startswith("adam_kernel") or "adam_kernel" in r[0]]
assert len(adam) >= 2, "need two optimiser steps to delimit one training step"
sel = int(os.environ.get("PROF_STEP", "-1"))
assert sel == -1 or 1 <= sel < len(adam), "PROF_STEP outside the run's %d optimiser launches" % len(adam)
a0, a1 = adam[sel - 1], adam[sel]
lo, hi = a0 + 1, a1 + 1
step = rows[lo:hi]
t0, t1 = rows[a0][3], rows[a1][3]
print("# step between %s adam_kernel ends: %.3f ms, %d kernels" % ("the last two" if sel == -1 else "the %d. and %d." % (sel, sel + 1), (t1 - t0) / 1e6, len(step)))
streams = {}
for r in step:
    streams.setdefault(r[1], []).append(r)
# the main stream is the one that carries the optimiser step (the side stream can hold more launches: chunked GEMMs, split passes)
main = rows[a1][1]
for sid, ks in sorted(streams.items(), key=lambda kv: -len(kv[1])):
    busy = sum(k[3] - k[2] for k in ks) / 1e3
    print("## stream %s%s: %d kernels, busy %.1f us" % (sid, " (main)" if sid == main else "", len(ks), busy))
    prev = t0 if sid == main else ks[0][2]
    gaps = 0.0
    for k in ks:
        gap = (k[2] - prev) / 1e3
        if sid == main and gap > 0:
            gaps += gap
        if (sid == main or show_all) and (gap >= min_gap or (k[3] - k[2]) / 1e3 >= 100):
            print("%10.1f  gap %7.1f  dur %8.1f  %s" % ((k[2] - t0) / 1e3, gap, (k[3] - k[2]) / 1e3, k[0].replace("(anonymous namespace)::", "")[:70]))
        prev = max(prev, k[3])
    if sid == main:
        print("## main-stream idle inside the step: %.1f us" % gaps)
