#!/usr/bin/env python3
"""Launch-boundness of small problems: wall time per ALS iteration against the sum of the hipEvent phases
(pbmc3k at k = 10, config 2).  usage: r4_small.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import singlet_amd as sa

def pbmc():
    g = np.load(os.path.join(ROOT, "tests", "golden", "pbmc3k_counts.npz"))
    p, dim = g["p"], g["dim"]
    i = g["di"].astype(np.int64)
    for c in range(dim[1]):
        i[p[c]:p[c + 1]] = np.cumsum(i[p[c]:p[c + 1]])
    return sa.dgCMatrix(g["x"].astype(np.float64), i.astype(np.int32), p, (int(dim[0]), int(dim[1])))

def rate(ctx, k, iters, label):
    ctx.fit_init(k, None)
    ctx.nmf_run(0.0, 5, 0.01, 0.01, 0.0, 0.0)
    ctx.fit_init(k, None)
    t0 = time.perf_counter()
    ctx.nmf_run(0.0, iters, 0.01, 0.01, 0.0, 0.0)
    wall = (time.perf_counter() - t0) / iters
    ctx.fit_init(k, None)
    ctx.timing_enable(True); ctx.timing_get(reset=True)
    ctx.nmf_run(0.0, iters, 0.01, 0.01, 0.0, 0.0)
    ph = ctx.timing_get(reset=True)
    ctx.timing_enable(False)
    busy = sum(v[0] for v in ph.values()) / iters
    print("%s k=%d: %.1f us per iteration wall, %.1f us in the phases (%s)" % (label, k, wall * 1e6, busy * 1e3,
          {n: round(v[0] / iters * 1e3, 1) for n, v in ph.items() if v[0]}))

c = sa.Context(0)
A = sa.PreprocessData(pbmc())
c.upload(A, None)
for k in (10, 30):
    rate(c, k, 200, "pbmc3k")
c.close()
c = sa.Context(0)
c.synth(20000, 50000, 20)
rate(c, 30, 100, "config 2")
c.close()
