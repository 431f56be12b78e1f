#!/usr/bin/env python3
"""Summarises rocprofv3 (rocpd sqlite) output directories into small text/CSV files for profiles/.
usage: python tools/rocpd_summary.py [--last N] <dir-with-results.db> [<more dirs>...] > summary.csv
--last N (r05, tools/profile_legs.py): per kernel name only its LAST N dispatches of the process (by start time) -- the timed
repetitions of one leg, without the launches that set the workload up."""
import glob
import sqlite3
import sys


LAST = 0
if len(sys.argv) > 2 and sys.argv[1] == "--last":
    LAST = int(sys.argv[2])
    del sys.argv[1:3]


def kernels(db):
    c = sqlite3.connect(db)
    src = "kernels"
    if LAST:
        src = "(select * from (select *, row_number() over (partition by name order by start desc) as rn from kernels) where rn <= %d)" % LAST
    q = ("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(scratch_size), "
         "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(grid_x), max(workgroup_x) from " + src + " "
         "group by name order by sum(duration) desc")
    return list(c.execute(q))


def counters(db):
    c = sqlite3.connect(db)
    try:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        if not cols:
            return []
        src = "counters_collection"
        if LAST:
            src = ("(select * from (select *, row_number() over (partition by kernel_name, counter_name order by start desc) as rn "
                   "from counters_collection) where rn <= %d)" % LAST)
        q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from " + src + " "
             "group by kernel_name, counter_name order by kernel_name")
        return list(c.execute(q))
    except sqlite3.Error:
        return []


for d in sys.argv[1:]:
    for db in sorted(glob.glob(d + "/**/*.db", recursive=True)):
        print("# %s" % db)
        print("kernel,calls,total_ms,avg_ms,min_ms,max_ms,scratch_B_per_lane,vgpr,agpr,sgpr,grid,wg")
        for r in kernels(db):
            name = r[0].split("(")[0].replace(",", ";")
            print("%s,%d,%.3f,%.3f,%.3f,%.3f,%s,%s,%s,%s,%s,%s" % (name, r[1], r[2] / 1e6, r[3] / 1e6, r[4] / 1e6, r[5] / 1e6,
                                                                  r[6], r[7], r[8], r[9], r[10], r[11]))
        cs = counters(db)
        if cs:
            print("kernel,counter,samples,avg_value,sum_value")
            for r in cs:
                print("%s,%s,%d,%.1f,%.1f" % (r[0].split("(")[0].replace(",", ";"), r[1], r[2], r[3], r[4]))
