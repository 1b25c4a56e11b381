#!/usr/bin/env python3
"""Time the masked (c_ard_nmf) loop on a resident synthetic shard: ms per iteration and per phase.
usage: ard_rate.py [cells] [genes] [k] [iters]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import singlet_amd as sa  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ctx = sa.Context(0)
ctx.synth(genes, cells, 20)
ctx.fit_init(k, None)
ctx.ard_run(0.0, 1, 0.01, 0.0, 123, 20, 1e9, 1)   # warm-up
ctx.fit_init(k, None)
ctx.timing_enable(True)
ctx.timing_get(reset=True)
ctx.sweeps_get(reset=True)
t0 = time.perf_counter()
r = ctx.ard_run(0.0, iters, 0.01, 0.0, 123, 20, 1e9, iters)
dt = time.perf_counter() - t0
ph = ctx.timing_get(reset=True)
sw = ctx.sweeps_get(reset=True)
print(json.dumps({"cells": cells, "genes": genes, "k": k, "iters": iters, "ms_per_iter": 1e3 * dt / iters,
                  "phases_ms_per_iter": {p: v[0] / iters for p, v in ph.items()}, "test_mse": r["test_mse"].tolist(),
                  "sweeps_per_column_per_iter": {"h": sw["h_sweeps"] / iters / cells, "w": sw["w_sweeps"] / iters / genes}}))
