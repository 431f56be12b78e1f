#!/usr/bin/env python3
"""rocprofv3 --kernel-trace (rocpd sqlite) -> the TIMELINE of the last N kernel dispatches of a process: start and end of every
dispatch relative to the first one, its queue and whether it overlapped another dispatch -- the evidence that two kernels really
ran side by side (tools/rocpd_summary.py only has durations).  usage: python tools/rocpd_timeline.py <dir-with-results.db> [N]"""
import glob
import sqlite3
import sys

n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for db in sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    sel = "name, start, end" + (", " + qcol if qcol else "")
    rows = list(c.execute("select %s from kernels order by start desc limit %d" % (sel, n)))[::-1]
    if not rows:
        continue
    t0 = rows[0][1]
    print("# %s: the last %d dispatches, ms relative to the first of them" % (db, len(rows)))
    print("kernel,start_ms,end_ms,duration_ms,queue,overlaps")
    for i, r in enumerate(rows):
        others = [o[0].split("(")[0].replace("void ", "").replace("tc::", "") for j, o in enumerate(rows) if j != i and o[1] < r[2] and r[1] < o[2]
                  and min(o[2], r[2]) - max(o[1], r[1]) > 20000]
        print("%s,%.3f,%.3f,%.3f,%s,%s" % (r[0].split("(")[0].replace("void ", "").replace(",", ";"), (r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6,
                                           r[3] if qcol else "", " + ".join(sorted(set(others)))))
