#!/usr/bin/env python3
"""BASELINE config 1 as a user would run it: the reference's bundled pbmc3k counts ->
PreprocessData (LogNormalize, on the device) -> run_nmf(rank = 10) -> project_model, all through
libsinglet_hip.so.  Needs an MI355X.

  python examples/pbmc3k_run_nmf.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import singlet_amd as sa  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "pbmc3k_counts.npz"))
p, dim = g["p"], g["dim"]
i = g["di"].astype(np.int64)          # row indices are stored as within-column deltas
for c in range(dim[1]):
    i[p[c]:p[c + 1]] = np.cumsum(i[p[c]:p[c + 1]])
counts = sa.dgCMatrix(g["x"].astype(np.float64), i.astype(np.int32), p, (int(dim[0]), int(dim[1])))
print("pbmc3k: %d genes x %d cells, %d non-zeros" % (counts.nrow, counts.ncol, counts.nnz))

A = sa.PreprocessData(counts)                         # Seurat::LogNormalize
t0 = time.perf_counter()
fit = sa.run_nmf(A, rank=10, tol=1e-4, maxit=100, verbose=False, L1=0.01, seed=123)
dt = time.perf_counter() - t0
print("run_nmf(rank=10): %d iterations in %.3f s, d = %s" % (fit["iter"], dt, np.round(fit["d"], 1)))
proj = sa.project_model(A, fit["w"])
print("project_model: h is %d x %d, relative change vs the fit's h: %.2e"
      % (proj["h"].shape[0], proj["h"].shape[1], np.linalg.norm(proj["h"] - fit["h"]) / np.linalg.norm(fit["h"])))
