#!/usr/bin/env python3
"""Time the dense front-end on a fully dense matrix: GEMM right-hand sides against the CSC-image path
(SGL_DENSE_GEMM=0).  usage: dense_rate.py [genes] [cells] [k] [iters]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import singlet_amd as sa  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
D = np.random.default_rng(1).random((m, n)) + 0.01
w0 = np.random.default_rng(2).random((m, k))
out = {"genes": m, "cells": n, "k": k, "iters": iters}
for mode in ("gemm", "csc_image"):
    os.environ["SGL_DENSE_GEMM"] = "1" if mode == "gemm" else "0"
    c = sa.Context(0)
    t0 = time.perf_counter()
    c.upload_dense(D)
    up = time.perf_counter() - t0
    c.fit_init(k, w0)
    c.nmf_run(0.0, 1, 0.01, 0.01, 0.0, 0.0)
    c.fit_init(k, w0)
    c.timing_enable(True)
    c.timing_get(reset=True)
    t0 = time.perf_counter()
    c.nmf_run(0.0, iters, 0.01, 0.01, 0.0, 0.0)
    dt = time.perf_counter() - t0
    ph = c.timing_get(reset=True)
    out[mode] = {"upload_s": up, "ms_per_iter": 1e3 * dt / iters, "phases_ms_per_iter": {p: v[0] / iters for p, v in ph.items() if v[0] > 0}}
    c.close()
print(json.dumps(out))
