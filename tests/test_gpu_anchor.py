"""The one external anchor the reference offers for this path (it holds no golden vectors): the 2022
render of its vignette, docs/articles/Guided_Clustering_with_NMF.html -- pbmc3k after the QC filter
(`nFeature_RNA > 200 & nFeature_RNA < 2500 & percent.mt < 5`, l.123-124) has 2638 cells (l.163),
RunNMF's automatic rank search ends at 15 factors (l.163, 174) and its cross-validation table shows test
errors of 0.131-0.136 (l.185).  R's random numbers are not reproducible here, so this is a sanity BOUND
on the order of magnitude (same data, same defaults, our engine), not a golden vector."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pbmc3k_qc(sa):
    g = np.load(os.path.join(GOLD, "pbmc3k_counts.npz"))
    p, di, x = g["p"].astype(np.int64), g["di"].astype(np.int64), g["x"].astype(np.float64)
    n = p.shape[0] - 1
    cs = np.cumsum(di)
    i = cs - np.repeat(cs[p[:-1]] - di[p[:-1]], np.diff(p))      # undo the per-column delta coding
    col = np.repeat(np.arange(n), np.diff(p))
    nfeature = np.diff(p)                                         # genes detected per cell
    total = np.bincount(col, weights=x, minlength=n)
    mt = np.isin(i, g["mt_rows"])
    pct_mt = 100.0 * np.bincount(col[mt], weights=x[mt], minlength=n) / total
    keep = (nfeature > 200) & (nfeature < 2500) & (pct_mt < 5)
    sel = keep[col]
    cnt = np.bincount(col[sel], minlength=n)[keep]
    A = sa.dgCMatrix(x[sel], i[sel].astype(np.int32), np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32),
                     (int(g["dim"][0]), int(keep.sum())))
    return A


@pytest.mark.timeout(900)
def test_pbmc3k_ard_nmf_lands_where_the_vignette_does(sa):
    counts = _pbmc3k_qc(sa)
    assert counts.ncol == 2638                                    # the vignette's cell count after QC
    A = sa.PreprocessData(counts)                                 # NormalizeData == LogNormalize, scale 1e4
    # RunNMF.Seurat's defaults (R/RunNMF.R:42-60 -> ard_nmf, l.128-144); k_max bounds the size of the random
    # initial w only (the reference draws 1e4 rows per replicate; the search never leaves the first few dozen)
    model = sa.ard_nmf(A, k_init=None, k_max=96, k_min=2, n_replicates=3, tol=1e-5, maxit=100, verbose=0, L1=0.01, L2=0,
                       threads=0, test_density=0.05, learning_rate=0.8, tol_overfit=1e-4, trace_test_mse=5, seed=123)
    cv = model["cv_data"]
    best = model["d"].shape[0]
    errs = np.array(cv.column("test_error"))
    final = {}
    for r in cv:                                                  # last traced error of every (k, rep)
        final[(r["k"], r["rep"])] = r["test_error"]
    at_best = [e for (k, rep), e in final.items() if abs(k - best) <= 2]
    print("pbmc3k ard_nmf: best rank", best, "ranks tried", sorted({r["k"] for r in cv}), "test error near the best rank",
          np.round(at_best, 4))
    assert 8 <= best <= 24, best                                  # the vignette: 15
    assert 0.10 < min(final.values()) < 0.17 and 0.10 < np.median(at_best) < 0.17   # the vignette: 0.131-0.136
    assert errs.min() > 0.05 and np.all(np.isfinite(errs))
    assert model["w"].shape == (13714, best) and model["h"].shape == (best, 2638)
    assert np.all(np.diff(model["d"]) <= 0)                       # sorted by d (R/run_nmf.R:65-68)
