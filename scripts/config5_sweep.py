#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU, down-scaled in cells: the rank sweep of ard_nmf / cross_validate_nmf
(k = 10, 20, ..., 100) on a resident synthetic shard.  For every rank: seconds per masked (c_ard_nmf)
iteration, seconds per plain (c_nmf) iteration, and the test-set error after `iters` iterations.
usage: config5_sweep.py [cells] [genes] [iters]   ->  one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import singlet_amd as sa  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = sa.Context(0)
ctx.synth(genes, cells, 20)
rows = []
for k in range(10, 101, 10):
    ctx.fit_init(k, None)
    ctx.nmf_run(0.0, 1, 0.01, 0.01, 0.0, 0.0)
    ctx.fit_init(k, None)
    t0 = time.perf_counter()
    ctx.nmf_run(0.0, iters, 0.01, 0.01, 0.0, 0.0)
    t_plain = (time.perf_counter() - t0) / iters
    ctx.fit_init(k, None)
    t0 = time.perf_counter()
    r = ctx.ard_run(0.0, iters, 0.01, 0.0, 123, 20, 1e9, iters)
    t_mask = (time.perf_counter() - t0 - 0.0) / iters
    rows.append({"k": k, "sec_per_plain_iter": t_plain, "sec_per_masked_iter": t_mask, "test_mse": float(r["test_mse"][-1])})
    print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps({"workload": "synthetic %d genes x %d cells, 5%% nnz, inv_density 20, %d iterations per fit" % (genes, cells, iters),
                  "ranks": rows}))
