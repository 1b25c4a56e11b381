#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE summaries of scripts/prof_r3.sh
(usage: make_traffic_json.py fetch.csv write.csv bench.json out.json; bench.json = the JSON line the profiled
bench run printed: its entry-stream layout is stored next to the bytes, and bench.py reports the bytes only
for a run with the same workload AND layout).

Units and correction as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
both counters are in KiB; FETCH_SIZE tallies each 128-byte request at 64 bytes -> x2; WRITE_SIZE as is.
Cross-check inside the same run: acc_tiled_reduce_kernel reads 13 slabs x 12 MB = 156 MB and writes
12 MB; corrected FETCH_SIZE = 2 x 76 183 KiB = 156.0 MB, WRITE_SIZE = 11 719 KiB = 12.0 MB."""
import csv
import json
import sys

fetch_csv, write_csv, bench_json, out = sys.argv[1:5]
layouts = json.loads(open(bench_json).read().strip().splitlines()[-1])["roofline"]["stream_layouts"]


def load(path, counter):
    rows = {}
    for r in csv.DictReader(open(path)):
        if r["counter"] == counter and "acc_tiled_kernel" in r["kernel"]:   # (round 4: a template, "void acc_tiled_kernel<3>")
            rows[("acc_tiled_kernel", int(r["grid_y"]))] = float(r["avg_per_dispatch"]) * 1024.0
    return rows


f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
res = {}
for (kern, gy), v in f.items():
    if kern != "acc_tiled_kernel":
        continue
    dom = "rhs_h" if gy == 1 else "rhs_w"
    res[dom] = int(2 * v + w[(kern, gy)])
    res[dom + "_detail"] = {"FETCH_SIZE_KiB_raw": v / 1024, "WRITE_SIZE_KiB_raw": w[(kern, gy)] / 1024,
                            "fetch_bytes_corrected_x2": int(2 * v), "write_bytes": int(w[(kern, gy)])}
json.dump({"workload": {"genes": 30000, "cells": 1000000, "k": 50, "inv_density": 20},
           "layout": layouts,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline`, scripts/prof_r6.sh; per launch of acc_tiled_kernel, averaged over its dispatches",
           "bytes_per_launch": res}, open(out, "w"), indent=1)
print(open(out).read())
