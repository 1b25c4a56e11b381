"""Host-side mirror of singlet's R interface for the ALS hot path.

The reference's host language is R (absent from this image), so the functions
the R drivers call are mirrored here one-for-one in Python, with the same
names, argument order, defaults and return fields:

  c_nmf / c_ard_nmf / c_project_model   R/RcppExports.R:28-30, 78-80, 24-26
  run_nmf                               R/run_nmf.R:18-77
  ard_nmf                               R/ard_nmf.R:31-193
  cross_validate_nmf                    R/cross_validate_nmf.R:18-105
  GetBestRank                           R/GetBestRank.R:8-46
  project_model                         R/ProjectData.R:11-19

Matrices follow R's orientation: w is returned k x m by c_nmf and m x k by
run_nmf / ard_nmf (they transpose and sort by d, R/run_nmf.R:65-68); h is
k x n.  All numerics run in libsinglet_hip.so on the GPU; nothing here computes
on the CPU beyond bookkeeping.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _lib
from ._lib import check, f64p, i32p, ptr
from .context import make_callbacks
from .sparse import as_dgCMatrix, dgCMatrix


# ---------------------------------------------------------------------------
# RcppExports layer
# ---------------------------------------------------------------------------
def _w_in(w, nrow):
    """R matrix k x m (column-major) -> (m, k) C-contiguous buffer."""
    w = np.asarray(w, dtype=np.float64)
    if w.ndim != 2 or w.shape[1] != nrow:
        raise ValueError("w must be a k x nrow(A) matrix (got %r, nrow(A) = %d)" % (w.shape, nrow))
    return np.ascontiguousarray(w.T)


def _verbose_log(verbose, ard=False):
    if not verbose:
        return None
    if ard:
        print("\n%4s | %8s | %8s \n---------------------------" % ("iter", "tol", "overfit"))

        def log(it, tol, of):
            print("%4d | %8.2e | %s" % (it, tol, ("%8s" % "-") if math.isnan(of) else ("%8.2e" % of)))
    else:
        print("\n%4s | %8s \n---------------" % ("iter", "tol"))

        def log(it, tol, of):
            print("%4d | %8.2e" % (it, tol))
    return log


CALL_TIMES_KEYS = ("h2d_s", "validate_s", "transpose_s", "fit_init_s", "iterate_s", "d2h_s", "h2d_bytes", "cached", "total_s")


def call_times():
    """Host wall-clock split of this thread's last one-shot call (c_nmf / c_ard_nmf): sgl_call_times_get."""
    out = np.zeros(len(CALL_TIMES_KEYS))
    check(_lib.load().sgl_call_times_get(ptr(out, f64p), int(out.size)))
    return dict(zip(CALL_TIMES_KEYS, (float(v) for v in out)))


def c_nmf(A, At, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w):
    """.Call(`_singlet_c_nmf`, ...) -> list(w = k x m, d = k, h = k x n)  (src/singlet.cpp:665)."""
    L = _lib.load()
    A = as_dgCMatrix(A)
    At = None if At is None else as_dgCMatrix(At)
    wb = _w_in(w, A.nrow)
    m, k = wb.shape
    n = A.ncol
    w_out, h_out, d_out = np.empty((m, k)), np.empty((n, k)), np.empty(k)
    n_iter = C.c_int32()
    tr = np.zeros(max(int(maxit), 1))
    cb = make_callbacks(_verbose_log(verbose))
    t = (ptr(At.x, f64p), ptr(At.i, i32p), ptr(At.p, i32p)) if At is not None else (None, None, None)
    check(L.sgl_c_nmf(ptr(A.x, f64p), ptr(A.i, i32p), ptr(A.p, i32p), *t, A.nrow, A.ncol, float(tol), int(maxit),
                      int(bool(verbose)), L1_w, L1_h, L2_w, L2_h, int(threads), ptr(wb, f64p), k, ptr(w_out, f64p),
                      ptr(d_out, f64p), ptr(h_out, f64p), C.byref(n_iter), ptr(tr, f64p), C.byref(cb)))
    return {"w": w_out.T, "d": d_out, "h": h_out.T, "iter": n_iter.value, "tol": tr[:n_iter.value].copy()}


def c_nmf_dense(A, At, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w):
    """.Call(`_singlet_c_nmf_dense`, ...) -> list(w, d, h)  (src/singlet.cpp:1052-1054).  A: dense m x n
    array; At is accepted for signature parity and ignored (the transpose is built on the device)."""
    L = _lib.load()
    A = np.asarray(A, dtype=np.float64)
    if A.ndim != 2:
        raise ValueError("A must be a matrix")
    m, n = A.shape
    Af = np.ascontiguousarray(A.T)   # column-major image
    wb = _w_in(w, m)
    k = wb.shape[1]
    w_out, h_out, d_out = np.empty((m, k)), np.empty((n, k)), np.empty(k)
    n_iter = C.c_int32()
    tr = np.zeros(max(int(maxit), 1))
    cb = make_callbacks(_verbose_log(verbose))
    check(L.sgl_c_nmf_dense(ptr(Af, f64p), m, n, float(tol), int(maxit), int(bool(verbose)), L1_w, L1_h, L2_w, L2_h,
                            int(threads), ptr(wb, f64p), k, ptr(w_out, f64p), ptr(d_out, f64p), ptr(h_out, f64p),
                            C.byref(n_iter), ptr(tr, f64p), C.byref(cb)))
    return {"w": w_out.T, "d": d_out, "h": h_out.T, "iter": n_iter.value, "tol": tr[:n_iter.value].copy()}


def _chunk_list(chunks):
    """ctypes image of a list of dgCMatrix column chunks: (n, x**, i**, p**, ncol*) and what must stay alive"""
    chunks = [as_dgCMatrix(a) for a in chunks]
    n = len(chunks)
    xs = (f64p * n)(*[ptr(a.x, f64p) for a in chunks])
    is_ = (i32p * n)(*[ptr(a.i, i32p) for a in chunks])
    ps = (i32p * n)(*[ptr(a.p, i32p) for a in chunks])
    nc = np.array([a.ncol for a in chunks], dtype=np.int32)
    return (n, xs, is_, ps, ptr(nc, i32p)), (chunks, xs, is_, ps, nc)


def _list_args(A_, At_):
    A_ = list(A_)
    if not A_:
        raise ValueError("A_ must hold at least one matrix")
    a, keep_a = _chunk_list(A_)
    nrow = keep_a[0][0].nrow
    if any(c.nrow != nrow for c in keep_a[0]):
        raise ValueError("all chunks of A_ must have the same number of rows")
    if At_ is None or len(At_) == 0:
        t, keep_t = (0, None, None, None, None), None
    else:
        t, keep_t = _chunk_list(At_)
    n = sum(c.ncol for c in keep_a[0])
    return a, t, nrow, n, (keep_a, keep_t)


def c_nmf_sparse_list(A_, At_, tol, maxit, verbose, L1, L2, threads, w):
    """.Call(`_singlet_c_nmf_sparse_list`, ...)  (src/singlet.cpp:715-743): A_ is a list of column chunks of A
    (the predict over chunks carries a running column offset, :384-402), At_ a list of column chunks of t(A)
    (None / empty: the transpose is built on the device).  The chunks are joined on the device."""
    L = _lib.load()
    a, t, nrow, n, keep = _list_args(A_, At_)
    wb = _w_in(w, nrow)
    m, k = wb.shape
    w_out, h_out, d_out = np.empty((m, k)), np.empty((n, k)), np.empty(k)
    n_iter = C.c_int32()
    tr = np.zeros(max(int(maxit), 1))
    cb = make_callbacks(_verbose_log(verbose))
    check(L.sgl_c_nmf_sparse_list(*a, *t, nrow, float(tol), int(maxit), int(bool(verbose)), L1, L2, int(threads), ptr(wb, f64p), k,
                                  ptr(w_out, f64p), ptr(d_out, f64p), ptr(h_out, f64p), C.byref(n_iter), ptr(tr, f64p),
                                  C.byref(cb)))
    return {"w": w_out.T, "d": d_out, "h": h_out.T, "iter": n_iter.value, "tol": tr[:n_iter.value].copy()}


def c_ard_nmf_sparse_list(A_, At_, tol, maxit, verbose, L1, L2, threads, w, rng_seed, inv_density, overfit_threshold,
                          trace_test_mse):
    """.Call(`_singlet_c_ard_nmf_sparse_list`, ...)  (src/singlet.cpp:1162-1234)."""
    L = _lib.load()
    a, t, nrow, n, keep = _list_args(A_, At_)
    wb = _w_in(w, nrow)
    m, k = wb.shape
    w_out, h_out, d_out = np.empty((m, k)), np.empty((n, k)), np.empty(k)
    cap = int(maxit) + 2
    tm, ft, so = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    itv = np.zeros(cap, dtype=np.int32)
    nt = C.c_int32()
    cb = make_callbacks(_verbose_log(verbose, ard=True))
    check(L.sgl_c_ard_nmf_sparse_list(*a, *t, nrow, float(tol), int(maxit), int(bool(verbose)), L1, L2, int(threads),
                                      ptr(wb, f64p), k, int(rng_seed), int(inv_density), float(overfit_threshold),
                                      int(trace_test_mse), ptr(w_out, f64p), ptr(d_out, f64p), ptr(h_out, f64p), ptr(tm, f64p),
                                      ptr(itv, i32p), ptr(ft, f64p), ptr(so, f64p), C.byref(nt), C.byref(cb)))
    q = nt.value
    return {"w": w_out.T, "d": d_out, "h": h_out.T, "test_mse": tm[:q].copy(), "iter": itv[:q].copy(),
            "tol": ft[:q].copy(), "score_overfit": so[:q].copy()}


def c_ard_nmf_dense(A, At, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density, overfit_threshold, trace_test_mse):
    """.Call(`_singlet_c_ard_nmf_dense`, ...)  (src/singlet.cpp:1357-1361).  A: dense m x n array; At is accepted
    for signature parity and ignored."""
    L = _lib.load()
    A = np.asarray(A, dtype=np.float64)
    if A.ndim != 2:
        raise ValueError("A must be a matrix")
    m, n = A.shape
    Af = np.ascontiguousarray(A.T)
    wb = _w_in(w, m)
    k = wb.shape[1]
    w_out, h_out, d_out = np.empty((m, k)), np.empty((n, k)), np.empty(k)
    cap = int(maxit) + 2
    tm, ft, so = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    itv = np.zeros(cap, dtype=np.int32)
    nt = C.c_int32()
    cb = make_callbacks(_verbose_log(verbose, ard=True))
    check(L.sgl_c_ard_nmf_dense(ptr(Af, f64p), m, n, float(tol), int(maxit), int(bool(verbose)), L1, L2, int(threads),
                                ptr(wb, f64p), k, int(seed), int(inv_density), float(overfit_threshold), int(trace_test_mse),
                                ptr(w_out, f64p), ptr(d_out, f64p), ptr(h_out, f64p), ptr(tm, f64p), ptr(itv, i32p),
                                ptr(ft, f64p), ptr(so, f64p), C.byref(nt), C.byref(cb)))
    q = nt.value
    return {"w": w_out.T, "d": d_out, "h": h_out.T, "test_mse": tm[:q].copy(), "iter": itv[:q].copy(),
            "tol": ft[:q].copy(), "score_overfit": so[:q].copy()}


def c_linked_nmf(A, At, tol, maxit, verbose, L1, L2, threads, w, link_h, link_w):
    """.Call(`_singlet_c_linked_nmf`, ...) -> list(w, d, h)  (src/singlet.cpp:1059-1086).  link_h / link_w
    are R matrices (rows x cols); a link whose column count does not match its side is ignored, as in
    the reference (R/RunLNMF.R passes a 1 x 1 matrix to switch a side off)."""
    L = _lib.load()
    A = as_dgCMatrix(A)
    At = None if At is None else as_dgCMatrix(At)
    wb = _w_in(w, A.nrow)
    m, k = wb.shape
    n = A.ncol
    w_out, h_out, d_out = np.empty((m, k)), np.empty((n, k)), np.empty(k)
    n_iter = C.c_int32()
    tr = np.zeros(max(int(maxit), 1))
    cb = make_callbacks(_verbose_log(verbose))
    t = (ptr(At.x, f64p), ptr(At.i, i32p), ptr(At.p, i32p)) if At is not None else (None, None, None)

    def link(Lk):
        if Lk is None:
            return None, 0, 0, None
        Lk = np.asarray(Lk, dtype=np.float64)
        if Lk.ndim != 2:
            raise ValueError("link matrices must be 2-D")
        buf = np.ascontiguousarray(Lk.T)   # column-major image of the R matrix
        return ptr(buf, f64p), Lk.shape[0], Lk.shape[1], buf
    lh, lhr, lhc, keep_h = link(link_h)
    lw, lwr, lwc, keep_w = link(link_w)
    check(L.sgl_c_linked_nmf(ptr(A.x, f64p), ptr(A.i, i32p), ptr(A.p, i32p), *t, A.nrow, A.ncol, float(tol), int(maxit),
                             int(bool(verbose)), L1, L2, int(threads), ptr(wb, f64p), k, lh, lhr, lhc, lw, lwr, lwc,
                             ptr(w_out, f64p), ptr(d_out, f64p), ptr(h_out, f64p), C.byref(n_iter), ptr(tr, f64p),
                             C.byref(cb)))
    return {"w": w_out.T, "d": d_out, "h": h_out.T, "iter": n_iter.value, "tol": tr[:n_iter.value].copy()}


def c_ard_nmf(A, At, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density, overfit_threshold, trace_test_mse):
    """.Call(`_singlet_c_ard_nmf`, ...) -> list(w, d, h, test_mse, iter, tol, score_overfit) (src/singlet.cpp:1144-1151)."""
    L = _lib.load()
    A = as_dgCMatrix(A)
    At = None if At is None else as_dgCMatrix(At)
    wb = _w_in(w, A.nrow)
    m, k = wb.shape
    n = A.ncol
    w_out, h_out, d_out = np.empty((m, k)), np.empty((n, k)), np.empty(k)
    cap = int(maxit) + 2
    tm, ft, so = np.zeros(cap), np.zeros(cap), np.zeros(cap)
    itv = np.zeros(cap, dtype=np.int32)
    nt = C.c_int32()
    cb = make_callbacks(_verbose_log(verbose, ard=True))
    t = (ptr(At.x, f64p), ptr(At.i, i32p), ptr(At.p, i32p)) if At is not None else (None, None, None)
    check(L.sgl_c_ard_nmf(ptr(A.x, f64p), ptr(A.i, i32p), ptr(A.p, i32p), *t, A.nrow, A.ncol, float(tol), int(maxit),
                          int(bool(verbose)), L1, L2, int(threads), ptr(wb, f64p), k, int(seed), int(inv_density),
                          float(overfit_threshold), int(trace_test_mse), ptr(w_out, f64p), ptr(d_out, f64p),
                          ptr(h_out, f64p), ptr(tm, f64p), ptr(itv, i32p), ptr(ft, f64p), ptr(so, f64p), C.byref(nt),
                          C.byref(cb)))
    q = nt.value
    return {"w": w_out.T, "d": d_out, "h": h_out.T, "test_mse": tm[:q].copy(), "iter": itv[:q].copy(),
            "tol": ft[:q].copy(), "score_overfit": so[:q].copy()}


def c_project_model(A, w, L1, L2, threads):
    """.Call(`_singlet_c_project_model`, ...) -> list(h = k x n, d = k)  (src/singlet.cpp:405-413)."""
    L = _lib.load()
    A = as_dgCMatrix(A)
    w = np.asarray(w, dtype=np.float64)
    if w.ndim != 2:
        raise ValueError("w must be a matrix")
    w_rows, w_cols = w.shape
    wf = np.ascontiguousarray(w.T)  # column-major image of w
    k = w_cols if w_rows == A.nrow else w_rows
    h_out, d_out = np.empty((A.ncol, k)), np.empty(k)
    check(L.sgl_c_project_model(ptr(A.x, f64p), ptr(A.i, i32p), ptr(A.p, i32p), A.nrow, A.ncol, ptr(wf, f64p),
                                w_rows, w_cols, L1, L2, int(threads), ptr(h_out, f64p), ptr(d_out, f64p)))
    return {"h": h_out.T, "d": d_out}


def Rcpp_predict(A, w, L1, L2, threads):
    """.Call(`_singlet_Rcpp_predict`, ...) -> h (k x n)  (src/singlet.cpp:350-367)."""
    L = _lib.load()
    A = as_dgCMatrix(A)
    w = np.asarray(w, dtype=np.float64)
    if w.ndim != 2:
        raise ValueError("w must be a matrix")
    w_rows, w_cols = w.shape
    wf = np.ascontiguousarray(w.T)
    k = w_cols if (w_rows == A.nrow and w_cols != A.nrow) else w_rows
    h_out = np.empty((A.ncol, k))
    check(L.sgl_rcpp_predict(ptr(A.x, f64p), ptr(A.i, i32p), ptr(A.p, i32p), A.nrow, A.ncol, ptr(wf, f64p), w_rows, w_cols,
                             L1, L2, int(threads), ptr(h_out, f64p)))
    return h_out.T


# ---------------------------------------------------------------------------
# R drivers
# ---------------------------------------------------------------------------
def _pair(v):
    v = list(np.atleast_1d(v))
    return (float(v[0]), float(v[0])) if len(v) != 2 else (float(v[0]), float(v[1]))


def _rng(seed):
    return seed if isinstance(seed, np.random.Generator) else np.random.default_rng(seed)


def _sort_model(model, rn=None, cn=None):
    # sort_index <- order(model$d, decreasing = TRUE)   R/run_nmf.R:65-68
    idx = np.argsort(-model["d"], kind="stable")
    model["d"] = model["d"][idx]
    model["w"] = model["w"].T[:, idx]
    model["h"] = model["h"][idx, :]
    k = model["d"].shape[0]
    model["factor_names"] = ["NMF_%d" % (q + 1) for q in range(k)]
    model["rownames_w"] = rn
    model["colnames_h"] = cn
    return model


def run_nmf(A, rank, tol=1e-4, maxit=100, verbose=True, L1=0.01, L2=0, threads=0, seed=None):
    """R/run_nmf.R:18-77 (sparse, single-matrix branch).  `seed` replaces R's global RNG state
    (stats::runif, l.55): an int or numpy Generator."""
    dense_mode = isinstance(A, np.ndarray)   # R/run_nmf.R:41-46: a base matrix stays dense
    if not dense_mode:
        A = as_dgCMatrix(A)
        if verbose:
            print("running with sparse optimization")
    L1 = _pair(L1)
    L2 = _pair(L2)
    nrow = A.shape[0] if dense_mode else A.nrow
    # w_init <- matrix(stats::runif(nrow(A) * rank), rank, nrow(A))
    w_init = _rng(seed).random((nrow, rank)).T
    if dense_mode:
        model = c_nmf_dense(A, None, tol, maxit, bool(verbose), L1[0], L1[1], L2[0], L2[1], threads, w_init)
        return _sort_model(model, None, None)
    model = c_nmf(A, None, tol, maxit, bool(verbose), L1[0], L1[1], L2[0], L2[1], threads, w_init)
    return _sort_model(model, A.Dimnames[0], A.Dimnames[1])


def _staged(A, op):
    """Upload A, run a staging operator on the device, return the transformed dgCMatrix."""
    from .context import Context
    A = as_dgCMatrix(A)
    c = Context(0)
    try:
        c.upload(A, None)
        op(c)
        x, i, p = c.download(0)
    finally:
        c.close()
    return dgCMatrix(x, i, p.astype(np.int32), A.Dim, A.Dimnames)


def PreprocessData(A, scale_factor=10000.0):
    """PreprocessData.dgCMatrix (R/PreprocessData.R:34-39): Seurat::LogNormalize of a counts matrix,
    log1p(x / colSums * scale_factor), computed on the device; dimnames kept."""
    return _staged(A, lambda c: c.log_normalize(scale_factor))


def weight_by_split(A_, split_by, n_groups):
    """.Call(`_singlet_weight_by_split`, A_, split_by, n_groups)  (src/singlet.cpp:119-144):
    returns a rescaled copy; split_by is the 0-based group of every column (R/RunNMF.R:86)."""
    A = as_dgCMatrix(A_)
    sb = np.ascontiguousarray(split_by, dtype=np.int32)
    if sb.shape[0] != A.ncol:
        raise ValueError("split_by needs one entry per column of A")
    x = np.empty(A.nnz, dtype=np.float64)
    check(_lib.load().sgl_c_weight_by_split(ptr(A.x, f64p), ptr(A.i, i32p), ptr(A.p, i32p), A.nrow, A.ncol, ptr(sb, i32p),
                                            int(n_groups), ptr(x, f64p)))
    return dgCMatrix(x, A.i, A.p, A.Dim, A.Dimnames)


def project_model(A, w, L1=0.01, L2=0, threads=0):
    """R/ProjectData.R:11-19."""
    A = as_dgCMatrix(A)
    w = np.asarray(w)
    if w.shape[0] != A.nrow and w.shape[1] != A.nrow:
        raise ValueError("'w' must share a common edge with the rows of 'A'")
    return c_project_model(A, w, L1, L2, threads)


class _ResidentFits:
    """One matrix kept in HBM across many fits (include/singlet_hip.h section 2): what R's ard_nmf /
    cross_validate_nmf do by calling c_ard_nmf / c_nmf again and again on the same A (R/ard_nmf.R:95-160,
    R/cross_validate_nmf.R:69-97), without re-uploading, re-transposing and re-validating it per call.
    Same arguments and return lists as c_ard_nmf / c_nmf; results are identical to the one-shot calls."""

    def __init__(self, A, device=0):
        from .context import Context
        self.A = as_dgCMatrix(A)
        self.ctx = Context(device)
        try:
            self.ctx.upload(self.A, None)
        except Exception:
            self.ctx.close()
            raise

    def close(self):
        self.ctx.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def c_ard_nmf(self, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density, overfit_threshold, trace_test_mse):
        wb = _w_in(w, self.A.nrow)
        k = wb.shape[1]
        self.ctx.fit_init(k, wb)
        r = self.ctx.ard_run(float(tol), int(maxit), L1, L2, int(seed), int(inv_density), float(overfit_threshold),
                             int(trace_test_mse), log=_verbose_log(verbose, ard=True))
        W, d, H = self.ctx.get_factors()
        return {"w": W.T, "d": d, "h": H.T, "test_mse": r["test_mse"], "iter": r["iter"], "tol": r["tol"],
                "score_overfit": r["score_overfit"]}

    def c_nmf(self, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w):
        wb = _w_in(w, self.A.nrow)
        k = wb.shape[1]
        self.ctx.fit_init(k, wb)
        n_iter, tr = self.ctx.nmf_run(float(tol), int(maxit), L1_w, L1_h, L2_w, L2_h, log=_verbose_log(verbose))
        W, d, H = self.ctx.get_factors()
        return {"w": W.T, "d": d, "h": H.T, "iter": n_iter, "tol": tr}


class _OneShotFits:
    """The same interface through the one-shot entry points (every call uploads A again)."""

    def __init__(self, A):
        self.A = as_dgCMatrix(A)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def c_ard_nmf(self, *args):
        return c_ard_nmf(self.A, None, *args)

    def c_nmf(self, *args):
        return c_nmf(self.A, None, *args)


class _OneShotListFits:
    """R's list branch (R/ard_nmf.R:45-76, 109-110, 176-178; R/cross_validate_nmf.R:27-50, 76-77): A is a list of
    column chunks (dgCMatrix, same rows); every fit goes through c_ard_nmf_sparse_list / c_nmf_sparse_list.  R builds a
    "distributed transpose" At on the host first (a list of row-block transposes); here At_ = None: the library joins
    the chunks into one resident matrix with 64-bit column pointers and builds t(A) on the device."""

    def __init__(self, chunks):
        self.chunks = [as_dgCMatrix(a) for a in chunks]

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def c_ard_nmf(self, *args):
        return c_ard_nmf_sparse_list(self.chunks, None, *args)

    def c_nmf(self, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w):
        # c_nmf_sparse_list(A, At, tol, maxit, verbose > 2, L1, L2, threads, w_init_this)   R/ard_nmf.R:178
        return c_nmf_sparse_list(self.chunks, None, tol, maxit, verbose, L1_w, L2_w, threads, w)


class _OneShotDenseFits:
    """R's dense branch (class(A)[[1]] == "matrix": R/ard_nmf.R:79-86, 105-106, 172-173): c_ard_nmf_dense /
    c_nmf_dense on the dense matrix (every column is solved, all-zero ones included: src/singlet.cpp:370-381)."""

    def __init__(self, A):
        self.A = np.asarray(A, dtype=np.float64)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def c_ard_nmf(self, *args):
        return c_ard_nmf_dense(self.A, None, *args)

    def c_nmf(self, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w):
        # R/ard_nmf.R:173 calls c_nmf_dense(A, At, tol, maxit, verbose > 2, L1, L2, threads, w_init_this): nine arguments
        # for an eleven-argument wrapper (R/RcppExports.R: L1_w, L1_h, L2_w, L2_h) -- an error in R; mirrored as the call
        # the sparse branch makes (L1, L1, L2, L2), the only reading under which the dense branch returns a model
        return c_nmf_dense(self.A, None, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w)


def _classify_input(A):
    """-> ("list", chunks) | ("dense", array) | ("sparse", dgCMatrix): the three branches of R/ard_nmf.R:45-90 and
    R/cross_validate_nmf.R:27-63 (`"list" %in% class(A)`, `class(A)[[1]] == "matrix"`, everything else -> dgCMatrix)."""
    if isinstance(A, (list, tuple)):
        chunks = [as_dgCMatrix(a) for a in A]   # "you must provide a list of all 'dgCMatrix' objects"
        if not chunks:
            raise ValueError("A is an empty list")
        if len({c.nrow for c in chunks}) != 1:
            raise ValueError("number of rows in all provided 'A' matrices are not identical")
        rn0 = chunks[0].Dimnames[0]
        if rn0 is not None and any(c.Dimnames[0] is not None and list(c.Dimnames[0]) != list(rn0) for c in chunks[1:]):
            raise ValueError("rownames of all dgCMatrix objects in list must be identical")
        return "list", chunks
    if isinstance(A, np.ndarray) and A.ndim == 2:
        return "dense", A
    return "sparse", as_dgCMatrix(A)


class CVData(list):
    """cv_data rows: dicts with k, rep, test_error, iter, tol (+ overfit_score from ard_nmf);
    the column sets match R/ard_nmf.R:93,118 and R/cross_validate_nmf.R:90."""

    def columns(self):
        return list(self[0].keys()) if self else []

    def column(self, name):
        return [r[name] for r in self]


def GetBestRank(df, tol_overfit=1e-4):
    """R/GetBestRank.R:8-46, line for line."""
    df = list(df)
    best_ranks = []
    for replicate in sorted({r["rep"] for r in df}):
        df_rep = [r for r in df if r["rep"] == replicate]
        max_rank = max(r["k"] for r in df_rep) + 1
        seen = []
        for r in df_rep:
            if r["k"] not in seen:
                seen.append(r["k"])
        for rank in seen:
            if rank < max_rank:
                te = [r["test_error"] for r in df_rep if r["k"] == rank]
                if len(te) > 1:
                    v2 = te[1:]
                    v1 = te[:-1]
                    if len(v1) >= 2:
                        for pos in range(1, len(v1)):
                            if v1[pos] > v1[pos - 1]:
                                v1[pos] = v1[pos - 1]
                    if max([0.0] + [(b - a) / (b + a) for a, b in zip(v1, v2)]) > tol_overfit:
                        max_rank = rank
        df_rep = [r for r in df_rep if r["k"] < max_rank]
        if len(df_rep) == 0:
            best_ranks.append(2)
        elif len(df) == 1:
            best_ranks.append(df_rep[0]["k"])
        else:
            # group_by(rep, k) %>% slice(which.max(iter)): groups come out sorted by k
            last = {}
            for r in df_rep:
                cur = last.get(r["k"])
                if cur is None or r["iter"] > cur["iter"]:
                    last[r["k"]] = r
            rows = [last[kk] for kk in sorted(last)]
            errs = [r["test_error"] for r in rows]
            best_ranks.append(rows[errs.index(min(errs))]["k"])
    return int(math.floor(sum(best_ranks) / len(best_ranks)))


def ard_nmf(A, k_init=2, k_max=100, k_min=2, n_replicates=1, tol=1e-5, cv_tol=1e-4, maxit=100, verbose=1, L1=0.01,
            L2=0, threads=0, test_density=0.05, learning_rate=1, tol_overfit=1e-3, trace_test_mse=1, seed=None,
            resident=True):
    """R/ard_nmf.R:31-193: automatic rank search, then the final fit -- all three input branches: one dgCMatrix
    (:82-85), a list of dgCMatrix column chunks (:45-76 -> c_*_sparse_list), a dense matrix (:79-86 -> c_*_dense).
    resident = True keeps a single dgCMatrix in HBM across all fits of the search (False, and always for the list and
    dense branches: one-shot calls, as the R code makes them)."""
    if not L1 < 1:
        raise ValueError("L1 penalty must be strictly in the range (0, 1]")
    if k_init is None or (isinstance(k_init, float) and math.isnan(k_init)) or k_init < k_min:
        k_init = k_min
    if k_min < 2:
        raise ValueError("k_min cannot be less than 2")
    kind, A = _classify_input(A)
    if kind == "list":
        nrow = A[0].nrow
        rn = A[0].Dimnames[0]
        cns = [c.Dimnames[1] for c in A]
        cn = None if any(c is None for c in cns) else [name for c in cns for name in c]   # rownames(At[[1]]): all cells
        fits = _OneShotListFits(A)
    elif kind == "dense":
        nrow, rn, cn = A.shape[0], None, None
        fits = _OneShotDenseFits(A)
    else:
        nrow, (rn, cn) = A.nrow, A.Dimnames
        fits = _ResidentFits(A) if resident else _OneShotFits(A)
    if verbose > 0:
        print("running with dense optimization" if kind == "dense" else "running with sparse optimization")
    rng = _rng(seed)
    # w_init <- lapply(1:n_replicates, function(x) matrix(runif(nrow(A) * k_max), k_max, nrow(A)))
    w_init = [rng.random((nrow, k_max)).T for _ in range(n_replicates)]
    test_seed = int(rng.integers(1, 2 ** 31 - 1))  # abs(.Random.seed[[3]])
    inv_density = int(round(1 / test_density))
    df = CVData()
    try:
        return _ard_nmf_search(fits, (rn, cn), df, w_init, test_seed, inv_density, k_init, k_max, k_min, n_replicates, tol, cv_tol,
                               maxit, verbose, L1, L2, threads, learning_rate, tol_overfit, trace_test_mse)
    finally:
        fits.close()


def _ard_nmf_search(fits, dimnames, df, w_init, test_seed, inv_density, k_init, k_max, k_min, n_replicates, tol, cv_tol, maxit,
                    verbose, L1, L2, threads, learning_rate, tol_overfit, trace_test_mse):
    for curr_rep in range(1, n_replicates + 1):
        if verbose >= 1 and n_replicates > 1:
            print("\nREPLICATE ", curr_rep, "/", n_replicates)
        step_size = 1.0
        curr_rank = k_init
        while step_size >= 1 and curr_rank <= k_max and curr_rank >= k_min:
            if verbose > 0:
                print("k =", curr_rank, ", rep =", curr_rep)
            w_init_this = w_init[curr_rep - 1][:curr_rank, :]
            model = fits.c_ard_nmf(cv_tol, maxit, verbose > 2, L1, L2, threads, w_init_this, test_seed + curr_rep,
                                   inv_density, tol_overfit, trace_test_mse)
            overfit_score = float(model["score_overfit"][-1])
            for q in range(len(model["test_mse"])):
                df.append({"k": int(curr_rank), "rep": int(curr_rep), "test_error": float(model["test_mse"][q]),
                           "iter": int(model["iter"][q]), "tol": float(model["tol"][q]),
                           "overfit_score": overfit_score})
            if overfit_score >= tol_overfit:
                k_max = curr_rank
            df_rep = sorted([r for r in df if r["rep"] == curr_rep], key=lambda r: r["k"])
            best_rank = GetBestRank([r for r in df_rep if r["k"] < k_max])
            ks = sorted({r["k"] for r in df_rep})
            if best_rank not in ks:
                raise RuntimeError("argument is of length zero")  # what R's `if (rank_ind == ...)` does here
            rank_ind = ks.index(best_rank) + 1
            if rank_ind == len(ks):
                step_size = step_size * (1 + learning_rate)
                curr_rank = best_rank + int(math.floor(step_size))
            elif rank_ind == 1:
                if math.floor(step_size) < best_rank:
                    curr_rank = best_rank - int(math.floor(step_size))
                    step_size = step_size * (learning_rate + 1)
                else:
                    curr_rank = best_rank // 2
            else:
                next_lower_rank = ks[rank_ind - 2]
                next_higher_rank = ks[rank_ind]
                diff_lower = best_rank - next_lower_rank
                diff_higher = next_higher_rank - best_rank
                higher_option = best_rank + diff_higher // 2
                lower_option = best_rank - diff_lower // 2
                if diff_lower <= 1 and diff_higher <= 1:
                    break
                elif diff_lower >= diff_higher:
                    curr_rank = lower_option
                else:
                    curr_rank = higher_option
    best_rank = GetBestRank(df, tol_overfit)
    if verbose > 0:
        print("\nFitting final model at k =", best_rank)
    w_init_this = w_init[0][:best_rank, :]
    model = fits.c_nmf(tol, maxit, verbose > 2, L1, L1, L2, L2, threads, w_init_this)
    model["cv_data"] = df
    return _sort_model(model, dimnames[0], dimnames[1])


def _replica_devices(devices):
    """devices of the replica sweep: an explicit list, a count, or SINGLET_REPLICA_GPUS=N from the environment
    (None / unset: device 0 only)."""
    if devices is None:
        n = int(os.environ.get("SINGLET_REPLICA_GPUS", "1") or "1")
        return list(range(max(n, 1)))
    if isinstance(devices, (int, np.integer)):
        return list(range(max(int(devices), 1)))
    return [int(d) for d in devices]


def _run_grid_on_replicas(A, devices, jobs, run):
    """SURVEY.md 8(e) "rank-sweep alternative": the (rank, replicate) fits of a grid are independent, so every
    device keeps its OWN resident copy of A and pulls fits from a shared queue (largest rank first: the cost of a
    masked fit grows like k^2) -- no communication at all.  One host thread per device (the library calls release
    the GIL; a context is only ever used by its own thread).  jobs: list of argument tuples; run(fits, job) -> result.
    Results come back in job order and do not depend on which device ran which fit."""
    import threading
    order = sorted(range(len(jobs)), key=lambda q: -jobs[q][0])
    lock = threading.Lock()
    results = [None] * len(jobs)
    errors = []

    def worker(dev):
        try:
            with _ResidentFits(A, dev) as fits:
                while True:
                    with lock:
                        if errors or not order:
                            return
                        q = order.pop(0)
                    results[q] = run(fits, jobs[q])
        except BaseException as e:  # noqa: BLE001 -- re-raised on the calling thread
            with lock:
                errors.append(e)

    threads = [threading.Thread(target=worker, args=(d,), name="singlet-replica-%d" % i) for i, d in enumerate(devices)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results


def cross_validate_nmf(A, ranks, n_replicates=3, tol=1e-4, maxit=100, verbose=1, L1=0.01, L2=0, threads=0,
                       test_density=0.05, tol_overfit=1e-4, trace_test_mse=5, seed=None, resident=True, devices=None):
    """R/cross_validate_nmf.R:18-105 -> cv table with k, rep, test_error, iter, tol; one dgCMatrix, a list of dgCMatrix
    column chunks (:27-50 -> c_ard_nmf_sparse_list) or a dense matrix (:57-60 -> c_ard_nmf_dense).
    resident = True keeps a single dgCMatrix in HBM across the whole (rank, replicate) grid.  devices (a list, a count, or
    SINGLET_REPLICA_GPUS=N): deal the independent fits of the grid out over several GPUs, each with its own
    resident copy of A (BASELINE config 5 on one node); the table is the one-device table, row for row."""
    if L1 >= 1:
        raise ValueError("L1 penalty must be strictly in the range (0, 1]")
    kind, A = _classify_input(A)
    nrow = A[0].nrow if kind == "list" else (A.shape[0] if kind == "dense" else A.nrow)
    ranks = [int(r) for r in np.atleast_1d(ranks)]
    rng = _rng(seed)
    w_init = [rng.random((nrow, max(ranks))).T for _ in range(n_replicates)]
    seeds = [int(rng.integers(1, 2 ** 31 - 1)) for _ in range(n_replicates)]  # abs(.Random.seed[[3 + rep]])
    inv_density = int(round(1 / test_density))
    df2 = CVData()
    grid = [(k, rep) for rep in range(1, n_replicates + 1) for k in ranks]  # expand.grid(k = ranks, rep = 1:n)

    def fit(fits, job):
        k, rep = job
        return fits.c_ard_nmf(tol, maxit, verbose > 1, L1, L2, threads, w_init[rep - 1][:k, :], seeds[rep - 1],
                              inv_density, tol_overfit, trace_test_mse)

    def rows(k, rep, model):
        for t in range(len(model["test_mse"])):
            df2.append({"k": k, "rep": rep, "test_error": float(model["test_mse"][t]), "iter": int(model["iter"][t]),
                        "tol": float(model["tol"][t])})

    devs = _replica_devices(devices)
    if kind != "sparse":
        fits = _OneShotListFits(A) if kind == "list" else _OneShotDenseFits(A)
        for k, rep in grid:
            rows(k, rep, fit(fits, (k, rep)))
        return df2
    if resident and len(devs) > 1:
        for (k, rep), model in zip(grid, _run_grid_on_replicas(A, devs, grid, fit)):
            rows(k, rep, model)
        return df2
    fits = _ResidentFits(A, devs[0]) if resident else _OneShotFits(A)
    try:
        for q, (k, rep) in enumerate(grid):
            if verbose > 1:
                print("k = %d, rep = %d (%d/%d):" % (k, rep, q + 1, len(grid)))
            model = fit(fits, (k, rep))
            rows(k, rep, model)
            if verbose > 1:
                print("test set error: %#.4e\n" % model["test_mse"][-1])
                if model["test_mse"][-1] / model["test_mse"][0] > (1 + tol_overfit):
                    print("overfitting detected, lower rank recommended")
    finally:
        fits.close()
    return df2
