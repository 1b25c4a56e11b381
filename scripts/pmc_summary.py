#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd sqlite output: per-kernel counter averages (pmc runs) or
kernel-time stats (kernel-trace runs).  usage: pmc_summary.py <results.db> [name-filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
rows = list(cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                        "group by kernel_name, counter_name"))
if rows:
    print("kernel,counter,avg_per_dispatch,dispatches")
    for k, c, v, n in rows:
        if flt in k:
            print('"%s",%s,%.6g,%d' % (k.split("(")[0], c, v, n))
else:
    print("kernel,calls,total_us,avg_us,pct")
    for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        if flt in r[0]:
            print('"%s",%d,%.1f,%.2f,%.2f' % (r[0].split("(")[0][:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
