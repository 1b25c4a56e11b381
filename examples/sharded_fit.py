#!/usr/bin/env python3
"""One fit on several GPUs (BASELINE config 4's shape, scaled by the arguments): the cells of a synthetic matrix sharded over
the devices of ONE process -- what an R session gets with SINGLET_NGPU=N -- through the library's own team (sgl_multi_*: RCCL
inside the library, one host thread per device).  On a box with fewer devices `--loopback` puts all ranks on device 0 (the
exchange is then a summing HIP kernel): the same team logic, no scaling.

  python examples/sharded_fit.py --gpus 8                      # 30 000 x 1 000 000, k = 50 on eight MI355X
  python examples/sharded_fit.py --gpus 8 --loopback --cells 200000
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import singlet_amd as sa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=2)
ap.add_argument("--loopback", action="store_true")
ap.add_argument("--genes", type=int, default=30000)
ap.add_argument("--cells", type=int, default=1000000)
ap.add_argument("--k", type=int, default=50)
ap.add_argument("--maxit", type=int, default=20)
ap.add_argument("--masked", action="store_true", help="c_ard_nmf (test set 1 / 20, traced every 5 iterations) instead of c_nmf")
a = ap.parse_args()

have = sa.device_count()
if not a.loopback and have < a.gpus:
    raise SystemExit("%d gfx950 device(s) visible, %d asked for (use --loopback to rehearse on one)" % (have, a.gpus))
with sa.Multi([0] * a.gpus if a.loopback else list(range(a.gpus))) as M:
    t0 = time.perf_counter()
    M.synth(a.genes, a.cells, 20)                       # every rank generates its own block of cells
    M.fit_init(a.k, None)
    print("%d ranks, %d genes x %d cells resident after %.1f s" % (a.gpus, a.genes, a.cells, time.perf_counter() - t0))
    t0 = time.perf_counter()
    if a.masked:
        r = M.ard_run(1e-5, a.maxit, 0.01, 0.0, 123, 20, 1e-3, 5)
        it, last = r["n_iter"], "test error %s" % np.round(r["test_mse"], 6)
    else:
        it, tols = M.nmf_run(1e-5, a.maxit, 0.01, 0.01, 0.0, 0.0)
        last = "tol %.3e" % tols[-1]
    dt = time.perf_counter() - t0
    W, d, H = M.get_factors()                           # w, d replicated on every rank; h gathered from the ranks' cell blocks
    print("%d iterations in %.2f s (%.1f ms each), %s; w %s, h %s, d[:4] = %s"
          % (it, dt, 1e3 * dt / max(it, 1), last, W.shape, H.shape, np.round(d[:4], 2)))
