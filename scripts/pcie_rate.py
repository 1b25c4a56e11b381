#!/usr/bin/env python3
"""PCIe-inclusive rate of the one-shot entry point (host dgCMatrix in, host factors out) next to the
resident-context rate, on config 2 (20 000 genes x 50 000 cells, 5 % nnz, k = 30).  DESIGN.md quotes it.
Uses the oracle only to GENERATE the host matrix (same hash generator as the device one)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import singlet_amd as sa  # noqa: E402
from oracle import oracle as ora  # noqa: E402

genes, cells, k, maxit = 20000, 50000, 30, 10
A = ora.synth_csc(genes, cells, 20)
At = A.t()
w0 = ora.synth_winit(k, genes)
dA, dAt = sa.dgCMatrix(A.x, A.i, A.p, (genes, cells)), sa.dgCMatrix(At.x, At.i, At.p, (cells, genes))
out = {"config": "20000 genes x 50000 cells, nnz %d, k=%d, maxit=%d" % (A.x.size, k, maxit)}
for name, at in (("one_shot_with_At", dAt), ("one_shot_At_built_on_device", None)):
    sa.c_nmf(dA, at, 0.0, 1, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)  # warm-up (library load, code objects)
    t0 = time.perf_counter()
    sa.c_nmf(dA, at, 0.0, maxit, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    t1 = time.perf_counter()
    sa.c_nmf(dA, at, 0.0, 1, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    t2 = time.perf_counter()
    out[name] = {"sec_%d_iters" % maxit: t1 - t0, "sec_1_iter": t2 - t1,
                 "iter_per_s_inclusive": maxit / (t1 - t0), "setup_sec_est": (t2 - t1) - ((t1 - t0) - (t2 - t1)) / (maxit - 1)}
ctx = sa.Context(0)
ctx.upload(dA, dAt)
ctx.fit_init(k, w0)
ctx.nmf_run(0.0, 2, 0.01, 0.01, 0.0, 0.0)
ctx.fit_init(k, w0)
t0 = time.perf_counter()
ctx.nmf_run(0.0, maxit, 0.01, 0.01, 0.0, 0.0)
t1 = time.perf_counter()
out["resident"] = {"sec_%d_iters" % maxit: t1 - t0, "iter_per_s": maxit / (t1 - t0)}
out["host_bytes_in"] = int(2 * (A.x.nbytes + A.i.nbytes) + A.p.nbytes + At.p.nbytes + w0.nbytes)
print(json.dumps(out))
