#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (results.db).

  pmc_summary.py <results.db> [name-filter]

--pmc runs: average counter value per dispatch, grouped by kernel AND launch grid (so the two
accumulate passes of an iteration, which use the same kernel on A and on t(A), stay apart).
--kernel-trace runs: calls / total / average duration per kernel and launch grid, in milliseconds
(the same numbers rocprofv3 --stats prints, kept per grid).
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()


def short(name):
    return name.split("(")[0][:90]


rows = list(cur.execute(
    "select kernel_name, grid_size_x, grid_size_y, counter_name, avg(value), count(*) from counters_collection "
    "group by kernel_name, grid_size_x, grid_size_y, counter_name"))
if rows:
    print("kernel,grid_x,grid_y,counter,avg_per_dispatch,dispatches")
    for k, gx, gy, c, v, n in rows:
        if flt in k:
            print('"%s",%d,%d,%s,%.6g,%d' % (short(k), gx, gy, c, v, n))
else:
    tot = list(cur.execute("select sum(end - start) from kernels"))[0][0] or 1
    print("kernel,grid_x,grid_y,calls,total_ms,avg_ms,min_ms,max_ms,pct_of_gpu_time")
    q = ("select name, grid_x, grid_y, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
         "from kernels group by name, grid_x, grid_y order by sum(end - start) desc")
    for name, gx, gy, n, s, a, mn, mx in cur.execute(q):
        if flt in name:
            print('"%s",%d,%d,%d,%.3f,%.4f,%.4f,%.4f,%.2f' % (short(name), gx, gy, n, s / 1e6, a / 1e6, mn / 1e6, mx / 1e6,
                                                           100.0 * s / tot))
