#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU, down-scaled in cells: the (rank, replicate) grid of cross_validate_nmf
(k = 10, 20, ..., 100; 3 replicates) on a RESIDENT synthetic shard -- the matrix is generated once and every
fit is sgl_fit_init + sgl_ard_run on the same context (what singlet_amd.cross_validate_nmf(resident=True)
does).  Per (k, rep): wall seconds of the fit, iterations run, seconds per masked iteration, last test error;
per rank also seconds per plain (c_nmf) iteration.
usage: config5_sweep.py [cells] [genes] [maxit] [trace_test_mse]   ->  one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import singlet_amd as sa  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
genes = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
maxit = int(sys.argv[3]) if len(sys.argv) > 3 else 10
trace = int(sys.argv[4]) if len(sys.argv) > 4 else 5
ctx = sa.Context(0)
t0 = time.perf_counter()
ctx.synth(genes, cells, 20)
gen_s = time.perf_counter() - t0
ctx.fit_init(10, None)
ctx.ard_run(0.0, 1, 0.01, 0.0, 1, 20, 1e9, 1)   # warm-up (module load, workspace)
rows, plain = [], {}
t_all = time.perf_counter()
for k in range(10, 101, 10):
    for rep in (1, 2, 3):
        t0 = time.perf_counter()
        ctx.fit_init(k, None, synth_seed=0x5EED + rep)          # a different initial w per replicate
        r = ctx.ard_run(1e-4, maxit, 0.01, 0.0, 1000 + rep, 20, 1e-4, trace)
        dt = time.perf_counter() - t0
        rows.append({"k": k, "rep": rep, "wall_s": dt, "iters": int(r["n_iter"]), "sec_per_masked_iter": dt / max(int(r["n_iter"]), 1),
                     "traces": len(r["test_mse"]), "test_mse": float(r["test_mse"][-1])})
        print(rows[-1], file=sys.stderr, flush=True)
total = time.perf_counter() - t_all
# seconds per plain (c_nmf) iteration at every rank, for the table in DESIGN.md: AFTER the grid and outside its wall time --
# cross_validate_nmf runs masked fits only (rounds 1 - 3 timed these iterations inside the grid loop: ~3 s of their figures)
t_plain = time.perf_counter()
for k in range(10, 101, 10):
    ctx.fit_init(k, None)
    ctx.nmf_run(0.0, 1, 0.01, 0.01, 0.0, 0.0)
    ctx.fit_init(k, None)
    t0 = time.perf_counter()
    ctx.nmf_run(0.0, 3, 0.01, 0.01, 0.0, 0.0)
    plain[k] = (time.perf_counter() - t0) / 3
t_plain = time.perf_counter() - t_plain
print(json.dumps({"workload": "synthetic %d genes x %d cells, 5%% nnz, inv_density 20, cv_tol 1e-4, maxit %d, trace_test_mse %d, "
                              "one resident context, fit set-up (entry streams, mask lists) included in wall_s" % (genes, cells, maxit, trace),
                  "generate_s": gen_s, "grid_wall_s": total, "grid_is": "the 30 masked fits, nothing else (the plain-iteration timing below ran after them)",
                  "plain_timing_wall_s": t_plain, "sec_per_plain_iter": plain, "fits": rows}))
