"""ctypes binding of oracle/libsinglet_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY -- see the header of singlet_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package `singlet_amd`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_SO selects another build of the same source (bench.py's -O3 -march=native timing variant)
_SO = os.environ.get("ORACLE_SO", os.path.join(_HERE, "libsinglet_oracle.so"))

_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    src = os.path.join(_HERE, "singlet_oracle.c")
    default = os.path.join(_HERE, "libsinglet_oracle.so")
    if force or not os.path.exists(default) or os.path.getmtime(default) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsinglet_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.ora_rng_rand.restype = C.c_uint64
        L.ora_rng_rand.argtypes = [C.c_uint64] * 3
        L.ora_rng_draw.restype = C.c_int
        L.ora_rng_draw.argtypes = [C.c_uint64] * 4
        L.ora_rng_mask.restype = None
        L.ora_rng_mask.argtypes = [C.c_uint64] * 5 + [_u8p]
        L.ora_cor.restype = C.c_double
        L.ora_cor.argtypes = [_f64p, _f64p, C.c_size_t]
        L.ora_aat.restype = None
        L.ora_aat.argtypes = [_f64p, C.c_int, C.c_int64, _f64p]
        L.ora_scale.restype = None
        L.ora_scale.argtypes = [_f64p, C.c_int, C.c_int64, _f64p]
        L.ora_nnls.restype = C.c_int
        L.ora_nnls.argtypes = [_f64p, _f64p, _f64p, C.c_int, C.c_double, C.c_double]
        csc = [_f64p, _i32p, _i32p]
        L.ora_predict.restype = None
        L.ora_predict.argtypes = csc + [C.c_int32, C.c_int32, _f64p, _f64p, C.c_int, C.c_double, C.c_double, C.c_int]
        L.ora_rhs.restype = None
        L.ora_rhs.argtypes = csc + [C.c_int32, C.c_int32, _f64p, _f64p, C.c_int]
        L.ora_predict_mask.restype = None
        L.ora_predict_mask.argtypes = csc + [C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, _f64p, _f64p, C.c_int,
                                             C.c_double, C.c_double, C.c_int, C.c_int]
        L.ora_predict_mask_off.restype = None
        L.ora_predict_mask_off.argtypes = csc + [C.c_int32, C.c_int32, C.c_uint64, C.c_uint64, _f64p, _f64p, C.c_int,
                                                 C.c_double, C.c_double, C.c_int, C.c_int, C.c_uint64, C.c_uint64]
        L.ora_synth_gene_count.restype = C.c_int64
        L.ora_synth_gene_count.argtypes = [C.c_uint64, C.c_uint64, _i64p, C.c_int64, C.c_int64, _i32p]
        L.ora_synth_gene_fill.restype = None
        L.ora_synth_gene_fill.argtypes = [C.c_uint64, C.c_uint64, _i64p, C.c_int64, C.c_int64, _f64p, _i32p, _i32p, _f64p]
        L.ora_mse_test.restype = C.c_double
        L.ora_mse_test.argtypes = csc + [C.c_int32, C.c_int32, _f64p, _f64p, _f64p, C.c_int, C.c_uint64, C.c_uint64,
                                         C.c_int]
        L.ora_c_nmf.restype = C.c_int
        L.ora_c_nmf.argtypes = csc + csc + [C.c_int32, C.c_int32, C.c_double, C.c_int] + [C.c_double] * 4 + [
            C.c_int, C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p, _i64p]
        L.ora_c_nmf_dense.restype = C.c_int
        L.ora_c_nmf_dense.argtypes = [_f64p, _f64p, C.c_int32, C.c_int32, C.c_double, C.c_int] + [C.c_double] * 4 + [
            C.c_int, C.c_int, _f64p, _f64p, _f64p, _f64p]
        L.ora_c_linked_nmf.restype = C.c_int
        L.ora_c_linked_nmf.argtypes = csc + csc + [C.c_int32, C.c_int32, C.c_double, C.c_int, C.c_double, C.c_double,
                                                   C.c_int, C.c_int, _f64p, _f64p, C.c_int32, C.c_int32, _f64p,
                                                   C.c_int32, C.c_int32, _f64p, _f64p, _f64p]
        L.ora_c_project_model.restype = C.c_int
        L.ora_c_project_model.argtypes = csc + [C.c_int32, C.c_int32, _f64p, C.c_int32, C.c_int32, C.c_double,
                                                C.c_double, C.c_int, _f64p, _f64p]
        L.ora_c_ard_nmf.restype = C.c_int
        L.ora_c_ard_nmf.argtypes = csc + csc + [C.c_int32, C.c_int32, C.c_double, C.c_int, C.c_double, C.c_double,
                                                C.c_int, C.c_int, _f64p, _f64p, _f64p, C.c_uint64, C.c_uint64,
                                                C.c_double, C.c_int, _f64p, _i32p, _f64p, _f64p, _i32p]
        pp_f, pp_i = C.POINTER(_f64p), C.POINTER(_i32p)
        lst = [C.c_int, pp_f, pp_i, pp_i, _i32p]
        L.ora_c_nmf_sparse_list.restype = C.c_int
        L.ora_c_nmf_sparse_list.argtypes = lst + lst + [C.c_int32, C.c_int32, C.c_double, C.c_int, C.c_double, C.c_double,
                                                        C.c_int, C.c_int, _f64p, _f64p, _f64p, _f64p]
        L.ora_c_ard_nmf_sparse_list.restype = C.c_int
        L.ora_c_ard_nmf_sparse_list.argtypes = lst + lst + [C.c_int32, C.c_int32, C.c_double, C.c_int, C.c_double, C.c_double,
                                                            C.c_int, C.c_int, _f64p, _f64p, _f64p, C.c_uint64, C.c_uint64,
                                                            C.c_double, C.c_int, _f64p, _i32p, _f64p, _f64p, _i32p]
        L.ora_c_ard_nmf_dense.restype = C.c_int
        L.ora_c_ard_nmf_dense.argtypes = [_f64p, _f64p, C.c_int32, C.c_int32, C.c_double, C.c_int, C.c_double, C.c_double,
                                          C.c_int, C.c_int, _f64p, _f64p, _f64p, C.c_uint64, C.c_uint64, C.c_double, C.c_int,
                                          _f64p, _i32p, _f64p, _f64p, _i32p]
        L.ora_log_normalize.restype = None
        L.ora_log_normalize.argtypes = [_f64p, _i32p, C.c_int32, C.c_double]
        L.ora_weight_by_split.restype = None
        L.ora_weight_by_split.argtypes = [_f64p, _i32p, C.c_int32, _i32p, C.c_int32]
        L.ora_synth_count.restype = C.c_int64
        L.ora_synth_count.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, _i32p]
        L.ora_synth_fill.restype = None
        L.ora_synth_fill.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_int64, _f64p, _i32p, _i32p,
                                     _f64p]
        L.ora_synth_winit.restype = None
        L.ora_synth_winit.argtypes = [C.c_uint64, C.c_int, C.c_int64, _f64p]
        L.ora_transpose.restype = None
        L.ora_transpose.argtypes = csc + [C.c_int32, C.c_int32, _f64p, _i32p, _i32p]
        L.ora_max_threads.restype = C.c_int
        L.ora_set_timing_skip.restype = None
        L.ora_set_timing_skip.argtypes = [C.c_int]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def _csc(x, i, p):
    x = np.ascontiguousarray(x, dtype=np.float64)
    i = np.ascontiguousarray(i, dtype=np.int32)
    p = np.ascontiguousarray(p, dtype=np.int32)
    return (x, i, p), (_p(x, _f64p), _p(i, _i32p), _p(p, _i32p))


class CSC:
    """dgCMatrix slots (inst/include/singlet.h:36-44): x, i, p, Dim."""

    def __init__(self, x, i, p, nrow, ncol):
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.i = np.ascontiguousarray(i, dtype=np.int32)
        self.p = np.ascontiguousarray(p, dtype=np.int32)
        self.nrow, self.ncol = int(nrow), int(ncol)

    @property
    def nnz(self):
        return int(self.p[-1])

    def t(self):
        return transpose(self)

    def to_dense(self):
        out = np.zeros((self.nrow, self.ncol))
        for c in range(self.ncol):
            s = slice(self.p[c], self.p[c + 1])
            out[self.i[s], c] = self.x[s]
        return out


SYNTH_SEED = 0x5EED
LEVELS16 = np.log1p(1.0 + np.arange(16, dtype=np.float64))


def rng_rand(state, i, j):
    return int(lib().ora_rng_rand(state, i, j))


def rng_mask(state, cell0, ncells, ngenes, inv_density):
    out = np.empty((ncells, ngenes), dtype=np.uint8)
    lib().ora_rng_mask(state, cell0, ncells, ngenes, inv_density, _p(out, _u8p))
    return out


def cor(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    return float(lib().ora_cor(_p(x, _f64p), _p(y, _f64p), x.size))


def aat(F):
    """F: (cols, k) C-contiguous == k x cols column-major."""
    F = np.ascontiguousarray(F, dtype=np.float64)
    cols, k = F.shape
    G = np.empty((k, k))
    lib().ora_aat(_p(F, _f64p), k, cols, _p(G, _f64p))
    return G


def scale(F):
    F = np.array(F, dtype=np.float64, order="C")
    cols, k = F.shape
    d = np.empty(k)
    lib().ora_scale(_p(F, _f64p), k, cols, _p(d, _f64p))
    return F, d


def nnls(a, b, x, L1=0.0, L2=0.0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.array(b, dtype=np.float64)
    x = np.array(x, dtype=np.float64)
    k = b.size
    it = lib().ora_nnls(_p(a, _f64p), _p(b, _f64p), _p(x, _f64p), k, L1, L2)
    return x, b, it


def rhs(A, F):
    keep, ptrs = _csc(A.x, A.i, A.p)
    F = np.ascontiguousarray(F, dtype=np.float64)
    k = F.shape[1]
    B = np.empty((A.ncol, k))
    lib().ora_rhs(*ptrs, A.nrow, A.ncol, _p(F, _f64p), _p(B, _f64p), k)
    return B


def predict(A, F, X, L1=0.0, L2=0.0, threads=0):
    """Matrices are passed as (cols, k) C-contiguous arrays (== k x cols col-major)."""
    keep, ptrs = _csc(A.x, A.i, A.p)
    F = np.ascontiguousarray(F, dtype=np.float64)
    X = np.array(X, dtype=np.float64, order="C")
    k = F.shape[1]
    lib().ora_predict(*ptrs, A.nrow, A.ncol, _p(F, _f64p), _p(X, _f64p), k, L1, L2, threads)
    return X


def predict_mask(A, seed, inv_density, F, X, L1=0.0, L2=0.0, threads=0, mask_t=False, col_offset=0, row_offset=0):
    """col_offset / row_offset: global index of A's first column / row in the mask hash -- A is then a slice of a
    larger matrix and is solved exactly as that matrix would solve these columns (the reference's `i + offset`, :485)."""
    keep, ptrs = _csc(A.x, A.i, A.p)
    F = np.ascontiguousarray(F, dtype=np.float64)
    X = np.array(X, dtype=np.float64, order="C")
    k = F.shape[1]
    if col_offset == 0 and row_offset == 0:
        lib().ora_predict_mask(*ptrs, A.nrow, A.ncol, seed, inv_density, _p(F, _f64p), _p(X, _f64p), k, L1, L2, threads,
                               int(mask_t))
    else:
        lib().ora_predict_mask_off(*ptrs, A.nrow, A.ncol, seed, inv_density, _p(F, _f64p), _p(X, _f64p), k, L1, L2, threads,
                                   int(mask_t), int(col_offset), int(row_offset))
    return X


def mse_test(A, w, d, h, seed, inv_density, threads=0):
    keep, ptrs = _csc(A.x, A.i, A.p)
    w = np.ascontiguousarray(w, dtype=np.float64)
    h = np.ascontiguousarray(h, dtype=np.float64)
    d = np.ascontiguousarray(d, dtype=np.float64)
    k = w.shape[1]
    return float(lib().ora_mse_test(*ptrs, A.nrow, A.ncol, _p(w, _f64p), _p(d, _f64p), _p(h, _f64p), k, seed,
                                    inv_density, threads))


def c_nmf(A, At, tol, maxit, L1_w, L1_h, L2_w, L2_h, threads, w, timing=False):
    """Mirror of c_nmf (src/singlet.cpp:669).  w: (m, k) array == k x m col-major.
    Returns dict(w (m,k), d (k), h (n,k), iter, tol (per iteration))."""
    ka, pa = _csc(A.x, A.i, A.p)
    kt, pt = _csc(At.x, At.i, At.p)
    w = np.array(w, dtype=np.float64, order="C")
    m, k = w.shape
    n = A.ncol
    assert m == A.nrow and At.nrow == n and At.ncol == m
    h = np.empty((n, k))
    d = np.empty(k)
    tr = np.zeros(max(maxit, 1))
    ph = np.zeros(4)
    sw = np.zeros(2, dtype=np.int64)
    it = lib().ora_c_nmf(*pa, *pt, m, n, tol, maxit, L1_w, L1_h, L2_w, L2_h, threads, k, _p(w, _f64p), _p(h, _f64p),
                         _p(d, _f64p), _p(tr, _f64p), _p(ph, _f64p), _p(sw, _i64p))
    out = dict(w=w, d=d, h=h, iter=it, tol=tr[:it].copy())
    if timing:
        out["phase_sec"] = ph
        out["sweeps"] = sw
    return out


def c_project_model(A, w, L1, L2, threads=0):
    """w: 2-D array in R orientation (either m x k or k x m), column-major semantics."""
    ka, pa = _csc(A.x, A.i, A.p)
    w = np.asarray(w, dtype=np.float64)
    w_rows, w_cols = w.shape
    wf = np.ascontiguousarray(w.T)  # C-order of transpose == column-major of w
    k = w_cols if w_rows == A.nrow else w_rows
    h = np.empty((A.ncol, k))
    d = np.empty(k)
    lib().ora_c_project_model(*pa, A.nrow, A.ncol, _p(wf, _f64p), w_rows, w_cols, L1, L2, threads, _p(h, _f64p),
                              _p(d, _f64p))
    return dict(h=h, d=d)


def c_nmf_dense(A, tol, maxit, L1_w, L1_h, L2_w, L2_h, threads, w):
    """c_nmf_dense (src/singlet.cpp:1052-1054).  A: dense (m, n) array; w: (m, k) (== k x m column-major)."""
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    Af = np.ascontiguousarray(A.T)      # column-major image of A
    Atf = np.ascontiguousarray(A)       # column-major image of t(A)
    w = np.array(w, dtype=np.float64, order="C")
    k = w.shape[1]
    h = np.empty((n, k))
    d = np.empty(k)
    tr = np.zeros(max(int(maxit), 1))
    it = lib().ora_c_nmf_dense(_p(Af, _f64p), _p(Atf, _f64p), m, n, tol, int(maxit), L1_w, L1_h, L2_w, L2_h, threads, k,
                               _p(w, _f64p), _p(h, _f64p), _p(d, _f64p), _p(tr, _f64p))
    return dict(w=w, d=d, h=h, iter=it, tol=tr[:it].copy())


def c_linked_nmf(A, At, tol, maxit, L1, L2, threads, w, link_h, link_w):
    """c_linked_nmf (src/singlet.cpp:1059-1086).  w: (m, k) array (== k x m column-major); link_h /
    link_w: R-orientation 2-D arrays (rows x cols) or None.  Returns w (m, k), d, h (n, k), iter, tol."""
    ka, pa = _csc(A.x, A.i, A.p)
    kt, pt = _csc(At.x, At.i, At.p)
    w = np.array(w, dtype=np.float64, order="C")
    m, k = w.shape
    n = A.ncol
    h = np.empty((n, k))
    d = np.empty(k)
    tr = np.zeros(max(int(maxit), 1))

    def link(Lk):
        if Lk is None:
            return None, 0, 0, None
        Lk = np.asarray(Lk, dtype=np.float64)
        buf = np.ascontiguousarray(Lk.T)   # column-major image
        return _p(buf, _f64p), Lk.shape[0], Lk.shape[1], buf
    lh, lhr, lhc, keep1 = link(link_h)
    lw, lwr, lwc, keep2 = link(link_w)
    it = lib().ora_c_linked_nmf(*pa, *pt, A.nrow, n, tol, int(maxit), L1, L2, threads, k, _p(w, _f64p), lh, lhr, lhc, lw,
                                lwr, lwc, _p(h, _f64p), _p(d, _f64p), _p(tr, _f64p))
    return dict(w=w, d=d, h=h, iter=it, tol=tr[:it].copy())


def rcpp_predict(A, w, L1, L2, threads=0):
    """Rcpp_predict (src/singlet.cpp:350-367): c_project_model without the two scale() calls.
    w in R orientation; transposed iff w.rows() == A.rows() && w.cols() != A.rows() (l.351).
    Returns h as an (n, k) array (== k x n column-major)."""
    w = np.asarray(w, dtype=np.float64)
    F = np.ascontiguousarray(w) if (w.shape[0] == A.nrow and w.shape[1] != A.nrow) else np.ascontiguousarray(w.T)
    return predict(A, F, np.zeros((A.ncol, F.shape[1])), L1, L2, threads)


def c_ard_nmf(A, At, tol, maxit, L1, L2, threads, w, seed, inv_density, overfit_threshold, trace_test_mse):
    ka, pa = _csc(A.x, A.i, A.p)
    kt, pt = _csc(At.x, At.i, At.p)
    w = np.array(w, dtype=np.float64, order="C")
    m, k = w.shape
    n = A.ncol
    h = np.empty((n, k))
    d = np.empty(k)
    cap = maxit + 2
    tm = np.zeros(cap)
    itv = np.zeros(cap, dtype=np.int32)
    ft = np.zeros(cap)
    so = np.zeros(cap)
    nt = C.c_int32(0)
    it = lib().ora_c_ard_nmf(*pa, *pt, m, n, tol, maxit, L1, L2, threads, k, _p(w, _f64p), _p(h, _f64p),
                             _p(d, _f64p), seed, inv_density, overfit_threshold, trace_test_mse, _p(tm, _f64p),
                             _p(itv, _i32p), _p(ft, _f64p), _p(so, _f64p), C.byref(nt))
    q = nt.value
    return dict(w=w, d=d, h=h, test_mse=tm[:q].copy(), iter=itv[:q].copy(), tol=ft[:q].copy(),
                score_overfit=so[:q].copy(), n_iter=it)


def _csc_list(chunks):
    """ctypes view of a list of CSC chunks: (count, x**, i**, p**, ncol*) + keep-alive objects"""
    n = len(chunks)
    xs = (_f64p * n)(*[_p(c.x, _f64p) for c in chunks])
    is_ = (_i32p * n)(*[_p(c.i, _i32p) for c in chunks])
    ps = (_i32p * n)(*[_p(c.p, _i32p) for c in chunks])
    nc = np.array([c.ncol for c in chunks], dtype=np.int32)
    return (n, xs, is_, ps, _p(nc, _i32p)), (xs, is_, ps, nc, chunks)


def c_nmf_sparse_list(A_, At_, tol, maxit, L1, L2, threads, w):
    """c_nmf_sparse_list (src/singlet.cpp:715-743): A_ column chunks of A, At_ column chunks of t(A)."""
    a, keep_a = _csc_list(A_)
    t, keep_t = _csc_list(At_)
    w = np.array(w, dtype=np.float64, order="C")
    m, k = w.shape
    n = At_[0].nrow
    h, d, tr = np.empty((n, k)), np.empty(k), np.zeros(max(int(maxit), 1))
    it = lib().ora_c_nmf_sparse_list(*a, *t, m, n, tol, int(maxit), L1, L2, threads, k, _p(w, _f64p), _p(h, _f64p),
                                     _p(d, _f64p), _p(tr, _f64p))
    return dict(w=w, d=d, h=h, iter=it, tol=tr[:it].copy())


def c_ard_nmf_sparse_list(A_, At_, tol, maxit, L1, L2, threads, w, seed, inv_density, overfit_threshold, trace_test_mse):
    """c_ard_nmf_sparse_list (src/singlet.cpp:1162-1234)."""
    a, keep_a = _csc_list(A_)
    t, keep_t = _csc_list(At_)
    w = np.array(w, dtype=np.float64, order="C")
    m, k = w.shape
    n = At_[0].nrow
    h, d = np.empty((n, k)), np.empty(k)
    cap = maxit + 2
    tm, ft, so, itv, nt = np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros(cap, dtype=np.int32), C.c_int32(0)
    it = lib().ora_c_ard_nmf_sparse_list(*a, *t, m, n, tol, int(maxit), L1, L2, threads, k, _p(w, _f64p), _p(h, _f64p),
                                         _p(d, _f64p), seed, inv_density, overfit_threshold, trace_test_mse, _p(tm, _f64p),
                                         _p(itv, _i32p), _p(ft, _f64p), _p(so, _f64p), C.byref(nt))
    q = nt.value
    return dict(w=w, d=d, h=h, test_mse=tm[:q].copy(), iter=itv[:q].copy(), tol=ft[:q].copy(), score_overfit=so[:q].copy(),
                n_iter=it)


def c_ard_nmf_dense(A, tol, maxit, L1, L2, threads, w, seed, inv_density, overfit_threshold, trace_test_mse):
    """c_ard_nmf_dense (src/singlet.cpp:1357-1361).  A: dense (m, n) array."""
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    Af, Atf = np.ascontiguousarray(A.T), np.ascontiguousarray(A)
    w = np.array(w, dtype=np.float64, order="C")
    k = w.shape[1]
    h, d = np.empty((n, k)), np.empty(k)
    cap = maxit + 2
    tm, ft, so, itv, nt = np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros(cap, dtype=np.int32), C.c_int32(0)
    it = lib().ora_c_ard_nmf_dense(_p(Af, _f64p), _p(Atf, _f64p), m, n, tol, int(maxit), L1, L2, threads, k, _p(w, _f64p),
                                   _p(h, _f64p), _p(d, _f64p), seed, inv_density, overfit_threshold, trace_test_mse,
                                   _p(tm, _f64p), _p(itv, _i32p), _p(ft, _f64p), _p(so, _f64p), C.byref(nt))
    q = nt.value
    return dict(w=w, d=d, h=h, test_mse=tm[:q].copy(), iter=itv[:q].copy(), tol=ft[:q].copy(), score_overfit=so[:q].copy(),
                n_iter=it)


def log_normalize(A, scale_factor=10000.0):
    """PreprocessData.dgCMatrix (R/PreprocessData.R:34-39): a new CSC with x <- log1p(x / colsum * scale)."""
    x = np.array(A.x, dtype=np.float64)
    p = np.ascontiguousarray(A.p, dtype=np.int32)
    lib().ora_log_normalize(_p(x, _f64p), _p(p, _i32p), A.ncol, float(scale_factor))
    return CSC(x, A.i, A.p, A.nrow, A.ncol)


def weight_by_split(A, split_by, n_groups):
    """weight_by_split (src/singlet.cpp:119-144): a new CSC."""
    x = np.array(A.x, dtype=np.float64)
    p = np.ascontiguousarray(A.p, dtype=np.int32)
    sb = np.ascontiguousarray(split_by, dtype=np.int32)
    lib().ora_weight_by_split(_p(x, _f64p), _p(p, _i32p), A.ncol, _p(sb, _i32p), int(n_groups))
    return CSC(x, A.i, A.p, A.nrow, A.ncol)


def transpose(A):
    ka, pa = _csc(A.x, A.i, A.p)
    nnz = A.nnz
    tx = np.empty(nnz)
    ti = np.empty(nnz, dtype=np.int32)
    tp = np.empty(A.nrow + 1, dtype=np.int32)
    lib().ora_transpose(*pa, A.nrow, A.ncol, _p(tx, _f64p), _p(ti, _i32p), _p(tp, _i32p))
    return CSC(tx, ti, tp, A.ncol, A.nrow)


def synth_csc(ngenes, ncells, inv_density=20, seed=SYNTH_SEED, cell0=0):
    """SURVEY 8(d) generator: genes x cells CSC for cells [cell0, cell0+ncells)."""
    p = np.empty(ncells + 1, dtype=np.int32)
    nnz = lib().ora_synth_count(seed, inv_density, cell0, ncells, ngenes, _p(p, _i32p))
    i = np.empty(nnz, dtype=np.int32)
    x = np.empty(nnz)
    lv = np.ascontiguousarray(LEVELS16)
    lib().ora_synth_fill(seed, inv_density, cell0, ncells, ngenes, _p(lv, _f64p), _p(p, _i32p), _p(i, _i32p),
                         _p(x, _f64p))
    return CSC(x, i, p, ngenes, ncells)


def synth_gene_columns(genes, ncells, inv_density=20, seed=SYNTH_SEED):
    """Columns `genes` of t(A) of the SURVEY 8(d) matrix over cells 0 .. ncells-1: CSC with ncells rows and one column
    per listed gene, without generating the whole matrix."""
    g = np.ascontiguousarray(genes, dtype=np.int64)
    p = np.empty(g.size + 1, dtype=np.int32)
    nnz = lib().ora_synth_gene_count(seed, inv_density, _p(g, _i64p), g.size, ncells, _p(p, _i32p))
    i = np.empty(nnz, dtype=np.int32)
    x = np.empty(nnz)
    lv = np.ascontiguousarray(LEVELS16)
    lib().ora_synth_gene_fill(seed, inv_density, _p(g, _i64p), g.size, ncells, _p(lv, _f64p), _p(p, _i32p), _p(i, _i32p),
                              _p(x, _f64p))
    return CSC(x, i, p, ncells, g.size)


def synth_winit(k, ngenes, seed=SYNTH_SEED):
    w = np.empty((ngenes, k))
    lib().ora_synth_winit(seed, k, ngenes, _p(w, _f64p))
    return w
