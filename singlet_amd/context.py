"""Device context: one resident shard of cells on one MI355X (sgl_ctx of
include/singlet_hip.h section 2).  Thin object wrapper over the C ABI."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, f64p, i32p, i64p, ptr, u64p, u8p

SYNTH_SEED = 0x5EED
# 16 value levels of the synthetic generator: log1p(1 + level)
LEVELS16 = np.log1p(1.0 + np.arange(16, dtype=np.float64))


def skew_weights16(sigma):
    """16 log-normal quantile levels exp(sigma * z_q), z_q the normal quantile of (q + 1/2) / 16, scaled to mean 1."""
    # normal quantiles of (q + 0.5) / 16, q = 0..15 (symmetric)
    z = np.array([-1.8627318674, -1.3180108973, -1.0099901692, -0.7764217611, -0.5791321623, -0.4022500653, -0.2372021093, -0.0784124127])
    z = np.concatenate([z, -z[::-1]])
    w = np.exp(float(sigma) * z)
    return w / w.mean()


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def make_callbacks(log=None, poll=None):
    """Builds an sgl_callbacks struct; keep the returned object alive during the call."""
    cb = _lib.Callbacks()
    keep = []
    if log is not None:
        fn = _lib.LOG_FN(lambda user, it, tol, of: log(it, tol, of))
        cb.log = fn
        keep.append(fn)
    if poll is not None:
        fn = _lib.POLL_FN(lambda user: int(bool(poll())))
        cb.poll = fn
        keep.append(fn)
    cb._keep = keep
    return cb


COMM_ID_BYTES = 128


def comm_unique_id():
    """RCCL unique id (bytes) for Context.comm_init_rank: rank 0 makes it, the host broadcasts it."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(_lib.load().sgl_comm_unique_id(buf))
    return buf.raw


def device_count():
    """gfx950 devices this process can use (sgl_device_count; 0 without the HIP runtime or a device)."""
    return int(_lib.load().sgl_device_count())


def comm_available():
    """(ok, path): can RCCL be bound in this process (no communication), and which library was opened.
    Ranks agree on this BEFORE comm_init_rank, which is collective and would otherwise hang on a rank
    that cannot join."""
    buf = C.create_string_buffer(512)
    rc = _lib.load().sgl_comm_available(buf, 512)
    return rc == 0, buf.value.decode("utf-8", "replace")


def split_cells_by_nnz(p, n):
    """The cell split of Multi.upload (sgl_multi_upload_csc): n + 1 boundaries of contiguous blocks of
    nearly equal non-zero count, each at least one cell."""
    p = np.ascontiguousarray(p, dtype=np.int32)
    lo = np.zeros(n + 1, dtype=np.int64)
    check(_lib.load().sgl_split_cells_by_nnz(ptr(p, i32p), int(p.shape[0] - 1), int(n), ptr(lo, i64p)))
    return lo


class Context:
    def __init__(self, device=0, _borrowed=None):
        self._L = _lib.load()
        if _borrowed is not None:   # a rank of a Multi: owned by it
            self._h = _borrowed
            self._owned = False
        else:
            h = C.c_void_p()
            check(self._L.sgl_create(int(device), C.byref(h)))
            self._h = h
            self._owned = True
        self._keep = []
        self.k = 0

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._L.sgl_destroy(self._h)
            self._h = None

    def comm_init_rank(self, nranks, rank, comm_id):
        """Join the native team of `nranks` processes (one per GPU) with the id rank 0 made
        (comm_unique_id); call before fit_init.  nmf_iterate / nmf_run then exchange over RCCL."""
        if len(comm_id) != COMM_ID_BYTES:
            raise ValueError("comm_id must be %d bytes" % COMM_ID_BYTES)
        buf = C.create_string_buffer(bytes(comm_id), COMM_ID_BYTES)
        check(self._L.sgl_comm_init_rank(self._h, int(nranks), int(rank), buf))

    def comm_info(self):
        """What the library's own communicator reports: ranks (ncclCommCount), whether it is RCCL, the library bound."""
        n, r = C.c_int32(), C.c_int32()
        buf = C.create_string_buffer(512)
        check(self._L.sgl_comm_info(self._h, C.byref(n), C.byref(r), buf, 512))
        return {"nranks": n.value, "is_rccl": bool(r.value), "path": buf.value.decode("utf-8", "replace")}

    def nmf_iterate(self, L1_w, L1_h, L2_w, L2_h):
        """One ALS iteration (any exchange mode); returns tol."""
        t = C.c_double()
        check(self._L.sgl_nmf_iterate(self._h, L1_w, L1_h, L2_w, L2_h, C.byref(t)))
        return t.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- matrix -----------------------------------------------------------
    def upload(self, A, At=None, cell_offset=0, ncells_total=0):
        a = (ptr(A.x, f64p), ptr(A.i, i32p), ptr(A.p, i32p))
        if At is not None:
            if At.Dim != (A.Dim[1], A.Dim[0]):
                raise ValueError("At must be the transpose of A")
            t = (ptr(At.x, f64p), ptr(At.i, i32p), ptr(At.p, i32p))
        else:
            t = (None, None, None)
        check(self._L.sgl_upload_csc(self._h, *a, *t, A.nrow, A.ncol, int(cell_offset), int(ncells_total)))
        self.k = 0

    def upload_dense(self, A):
        """A: dense (nrow, ncol) array (sgl_upload_dense: CSC image built on the device, GEMM right-hand sides when
        more than half of it is non-zero)."""
        A = np.asarray(A, dtype=np.float64)
        Af = np.ascontiguousarray(A.T)   # column-major image
        check(self._L.sgl_upload_dense(self._h, ptr(Af, f64p), A.shape[0], A.shape[1]))
        self.k = 0

    def synth(self, ngenes, ncells_local, inv_density=20, seed=SYNTH_SEED, cell_offset=0, ncells_total=0, skew=None):
        """skew = (sigma_cells, sigma_genes): the skewed generator (log-normal weights per cell and per gene)."""
        lv = _f(LEVELS16)
        if skew is None:
            check(self._L.sgl_synth_csc(self._h, seed, inv_density, ptr(lv, f64p), int(ngenes), int(cell_offset),
                                        int(ncells_local), int(ncells_total)))
        else:
            cw, gw = _f(skew_weights16(skew[0])), _f(skew_weights16(skew[1]))
            check(self._L.sgl_synth_csc_skewed(self._h, seed, inv_density, ptr(lv, f64p), int(ngenes), int(cell_offset),
                                               int(ncells_local), int(ncells_total), ptr(cw, f64p), ptr(gw, f64p)))
        self.k = 0

    def dims(self):
        nr, nc, nz = C.c_int32(), C.c_int32(), C.c_int64()
        check(self._L.sgl_dims(self._h, C.byref(nr), C.byref(nc), C.byref(nz)))
        return nr.value, nc.value, nz.value

    def download(self, which=0):
        nr, nc, nz = self.dims()
        ncol = nr if which else nc
        x = np.empty(nz)
        i = np.empty(nz, dtype=np.int32)
        p = np.empty(ncol + 1, dtype=np.int64)
        check(self._L.sgl_download_csc(self._h, int(which), ptr(x, f64p), ptr(i, i32p), ptr(p, i64p)))
        return x, i, p

    def col_counts(self, which=0):
        """Non-zeros per column of the resident A (which = 0: per cell) or t(A) (which = 1: per gene): the column
        pointers only (sgl_download_csc with NULL value / index buffers)."""
        nr, nc, _ = self.dims()
        p = np.empty((nr if which else nc) + 1, dtype=np.int64)
        check(self._L.sgl_download_csc(self._h, int(which), None, None, ptr(p, i64p)))
        return np.diff(p)

    # -- fit ----------------------------------------------------------------
    def log_normalize(self, scale_factor=10000.0):
        """Seurat::LogNormalize on the resident shard (R/PreprocessData.R:34-39)."""
        check(self._L.sgl_log_normalize(self._h, float(scale_factor)))
        self.k = 0

    def weight_by_split(self, split_by, n_groups):
        """weight_by_split (src/singlet.cpp:119-144) on the resident shard; split_by: 0-based group per local cell."""
        sb = np.ascontiguousarray(split_by, dtype=np.int32)
        _, nc, _ = self.dims()
        if sb.shape != (nc,):
            raise ValueError("split_by must have one entry per cell (column) of A")
        check(self._L.sgl_weight_by_split(self._h, ptr(sb, i32p), int(n_groups)))
        self.k = 0

    def fit_init(self, k, w_init=None, synth_seed=SYNTH_SEED):
        """w_init: (m, k) C-contiguous (== k x m column-major) or None for the synthetic init."""
        w = None
        if w_init is not None:
            w = _f(w_init)
            nr, _, _ = self.dims()
            if w.shape != (nr, k):
                raise ValueError("w_init must be k x nrow(A) (got %r for k=%d, nrow=%d)" % (w.shape[::-1], k, nr))
        check(self._L.sgl_fit_init(self._h, int(k), ptr(w, f64p), synth_seed))
        self.k = int(k)

    def set_stream(self, stream_ptr):
        check(self._L.sgl_set_stream(self._h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def set_allreduce(self, fn):
        """fn(dev_ptr: int, count: int) -> None; sums `count` doubles at dev_ptr over all shards."""
        if fn is None:
            cfn = C.cast(None, _lib.ALLREDUCE_FN)
        else:
            def tramp(user, dev_ptr, count):
                try:
                    fn(int(dev_ptr), int(count))
                    return 0
                except Exception:  # noqa: BLE001 - must not unwind through C
                    import traceback
                    traceback.print_exc()
                    return 1
            cfn = _lib.ALLREDUCE_FN(tramp)
        self._keep = [cfn]
        check(self._L.sgl_set_allreduce(self._h, cfn, None))

    def step_begin(self):
        check(self._L.sgl_step_begin(self._h))

    def step_h(self, L1, L2):
        check(self._L.sgl_step_h(self._h, L1, L2))

    def step_scale_h(self):
        check(self._L.sgl_step_scale_h(self._h))

    def step_w(self, L1, L2):
        check(self._L.sgl_step_w(self._h, L1, L2))

    def step_h_masked(self, L1, L2, seed, inv_density):
        """predict_mask(A, ...) on the resident fit: the masked H-update (unscaled h)."""
        check(self._L.sgl_step_h_masked(self._h, L1, L2, int(seed), int(inv_density)))

    def step_w_masked(self, L1, L2, seed, inv_density):
        """predict_mask(At, ..., mask_t = true) on the resident fit: the masked W-update (unscaled w)."""
        check(self._L.sgl_step_w_masked(self._h, L1, L2, int(seed), int(inv_density)))

    def step_scale_w(self):
        t = C.c_double()
        check(self._L.sgl_step_scale_w(self._h, C.byref(t)))
        return t.value

    def nmf_run(self, tol, maxit, L1_w, L1_h, L2_w, L2_h, log=None, poll=None):
        n_iter = C.c_int32()
        tr = np.zeros(max(int(maxit), 1))
        cb = make_callbacks(log, poll)
        check(self._L.sgl_nmf_run(self._h, tol, int(maxit), L1_w, L1_h, L2_w, L2_h, C.byref(n_iter), ptr(tr, f64p),
                                  C.byref(cb)))
        return n_iter.value, tr[:n_iter.value].copy()

    def ard_run(self, tol, maxit, L1, L2, seed, inv_density, overfit_threshold, trace_test_mse, log=None, poll=None):
        cap = int(maxit) + 2
        tm, ft, so = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        itv = np.zeros(cap, dtype=np.int32)
        nt, nit = C.c_int32(), C.c_int32()
        cb = make_callbacks(log, poll)
        check(self._L.sgl_ard_run(self._h, tol, int(maxit), L1, L2, int(seed), int(inv_density), overfit_threshold,
                                  int(trace_test_mse), ptr(tm, f64p), ptr(itv, i32p), ptr(ft, f64p), ptr(so, f64p),
                                  C.byref(nt), C.byref(nit), C.byref(cb)))
        q = nt.value
        return dict(test_mse=tm[:q].copy(), iter=itv[:q].copy(), tol=ft[:q].copy(), score_overfit=so[:q].copy(),
                    n_iter=nit.value)

    def project_run(self, L1, L2):
        check(self._L.sgl_project_run(self._h, L1, L2))

    def get_factors(self, w=True, d=True, h=True):
        nr, nc, _ = self.dims()
        k = self.k
        W = np.empty((nr, k)) if w else None
        D = np.empty(k) if d else None
        H = np.empty((nc, k)) if h else None
        check(self._L.sgl_get_factors(self._h, ptr(W, f64p), ptr(D, f64p), ptr(H, f64p)))
        return W, D, H

    def set_factors(self, w=None, d=None, h=None):
        w = None if w is None else _f(w)
        d = None if d is None else _f(d)
        h = None if h is None else _f(h)
        check(self._L.sgl_set_factors(self._h, ptr(w, f64p), ptr(d, f64p), ptr(h, f64p)))

    # -- single operators ---------------------------------------------------
    def op_rand(self, state, i, j):
        i = np.ascontiguousarray(i, dtype=np.uint64)
        j = np.ascontiguousarray(j, dtype=np.uint64)
        out = np.empty(i.shape, dtype=np.uint64)
        check(self._L.sgl_op_rand(self._h, int(state), ptr(i, u64p), ptr(j, u64p), i.size, ptr(out, u64p)))
        return out

    def op_mask(self, state, inv_density, cell0, ncells, ngenes):
        out = np.empty((ncells, ngenes), dtype=np.uint8)
        check(self._L.sgl_op_mask(self._h, int(state), int(inv_density), int(cell0), ncells, ngenes, ptr(out, u8p)))
        return out

    def op_gram(self, F):
        F = _f(F)
        cols, k = F.shape
        G = np.empty((k, k))
        check(self._L.sgl_op_gram(self._h, ptr(F, f64p), k, cols, ptr(G, f64p)))
        return G

    def op_rhs(self, which, F):
        F = _f(F)
        nr, nc, _ = self.dims()
        k = F.shape[1]
        ncol = nr if (which & 1) else nc   # which & 2 selects the LDS-tiled kernel
        nrow = nc if (which & 1) else nr
        if F.shape[0] != nrow:
            raise ValueError("F must have %d rows" % nrow)
        B = np.empty((ncol, k))
        check(self._L.sgl_op_rhs(self._h, int(which), ptr(F, f64p), k, ptr(B, f64p)))
        return B

    def op_nnls(self, G, B, X, L1=0.0, L2=0.0):
        G, B = _f(G), _f(B)
        X = np.array(X, dtype=np.float64, order="C")
        ncols, k = B.shape
        sw = C.c_int32()
        check(self._L.sgl_op_nnls(self._h, ptr(G, f64p), ptr(B, f64p), ptr(X, f64p), k, ncols, L1, L2, C.byref(sw)))
        return X, sw.value

    def op_mask_gram(self, F, G, ncols, seed, inv_density, mask_t=0, col_offset=0, row_offset=0, use_lists=False):
        """Per-column Gram downdates of predict_mask for columns 0 .. ncols-1: F is nrow x k, G k x k or None (raw sums)."""
        F = _f(F)
        nrow, k = F.shape
        out = np.empty((ncols, k, k))
        Gp = ptr(_f(G), f64p) if G is not None else None
        check(self._L.sgl_op_mask_gram(self._h, ptr(F, f64p), Gp, k, nrow, ncols, int(seed), int(inv_density), int(mask_t),
                                       int(col_offset), int(row_offset), 1 if use_lists else 0, ptr(out, f64p)))
        return out

    def op_scale(self, F):
        F = np.array(F, dtype=np.float64, order="C")
        cols, k = F.shape
        d = np.empty(k)
        check(self._L.sgl_op_scale(self._h, ptr(F, f64p), k, cols, ptr(d, f64p)))
        return F, d

    def op_cor(self, x, y):
        x, y = _f(x), _f(y)
        out = C.c_double()
        check(self._L.sgl_op_cor(self._h, ptr(x, f64p), ptr(y, f64p), x.size, C.byref(out)))
        return out.value

    def op_mse_test(self, seed, inv_density):
        out = C.c_double()
        check(self._L.sgl_op_mse_test(self._h, int(seed), int(inv_density), C.byref(out)))
        return out.value

    # -- timing -------------------------------------------------------------
    def timing_enable(self, on=True):
        check(self._L.sgl_timing_enable(self._h, int(bool(on))))

    def timing_get(self, reset=False):
        ms = np.zeros(_lib.SGL_PH_COUNT)
        calls = np.zeros(_lib.SGL_PH_COUNT, dtype=np.int64)
        check(self._L.sgl_timing_get(self._h, ptr(ms, f64p), ptr(calls, i64p), int(reset)))
        return {n: (float(ms[q]), int(calls[q])) for q, n in enumerate(_lib.SGL_PH_NAMES)}

    def sweeps_get(self, reset=False):
        out = np.zeros(4, dtype=np.int64)
        check(self._L.sgl_sweeps_get(self._h, ptr(out, i64p), int(reset)))
        return dict(h_sweeps=int(out[0]), w_sweeps=int(out[1]), h_wave_sweeps=int(out[2]), w_wave_sweeps=int(out[3]))

    def layout_get(self):
        """Entry-stream layout of the current fit: {'A': {...}, 'At': {...}} (all zero on the plain CSC path)."""
        out = np.zeros(10, dtype=np.int64)
        check(self._L.sgl_layout_get(self._h, ptr(out, i64p)))
        keys = ("entries", "tiles", "tile_rows", "tile_ranges", "col_blocks")
        return {name: dict(zip(keys, (int(v) for v in out[5 * o:5 * o + 5]))) for o, name in enumerate(("A", "At"))}

    def mask_pairs(self):
        """(pairs listed per cell, per gene) of the mask the current fit runs under (sgl_mask_pairs; 0 where no lists are built)."""
        out = np.zeros(2, dtype=np.int64)
        check(self._L.sgl_mask_pairs(self._h, ptr(out, i64p)))
        return int(out[0]), int(out[1])

    def layout_builds(self):
        """(stream of A, of At, mask lists of the cell side, of the gene side): times each has been written on this context
        (a re-init at an unchanged rank adds no stream, a masked fit under a recently used seed no lists)."""
        out = np.zeros(4, dtype=np.int64)
        check(self._L.sgl_layout_builds(self._h, ptr(out, i64p)))
        return tuple(int(v) for v in out)


class Multi:
    """sgl_multi: ONE process driving several devices, cells sharded, exchange over RCCL inside the
    library (include/singlet_hip.h section 2b).  devices: list of device ids -- all distinct (RCCL) or
    all equal (ranks share one device and exchange through a HIP kernel: the test configuration)."""

    def __init__(self, devices):
        self._L = _lib.load()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        check(self._L.sgl_multi_create(int(dev.size), ptr(dev, i32p), C.byref(h)))
        self._h = h
        self.n = int(dev.size)
        self.k = 0
        self._dims = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.sgl_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def rank_ctx(self, r):
        h = C.c_void_p()
        check(self._L.sgl_multi_ctx(self._h, int(r), C.byref(h)))
        c = Context(_borrowed=h)
        c.k = self.k
        return c

    def upload(self, A):
        check(self._L.sgl_multi_upload_csc(self._h, ptr(A.x, f64p), ptr(A.i, i32p), ptr(A.p, i32p), A.nrow, A.ncol))
        self._dims = (A.nrow, A.ncol)
        self.k = 0

    def synth(self, ngenes, ncells_total, inv_density=20, seed=SYNTH_SEED):
        lv = _f(LEVELS16)
        check(self._L.sgl_multi_synth_csc(self._h, seed, inv_density, ptr(lv, f64p), int(ngenes), int(ncells_total)))
        self._dims = (int(ngenes), int(ncells_total))
        self.k = 0

    def fit_init(self, k, w_init=None, synth_seed=SYNTH_SEED):
        w = None
        if w_init is not None:
            w = _f(w_init)
            if w.shape != (self._dims[0], k):
                raise ValueError("w_init must be k x nrow(A)")
        check(self._L.sgl_multi_fit_init(self._h, int(k), ptr(w, f64p), synth_seed))
        self.k = int(k)

    def set_links(self, link_h=None, link_w=None):
        """c_linked_nmf's link matrices (rows x cols, as R holds them) for the whole matrix; call after fit_init."""
        def img(Lk):
            if Lk is None:
                return None, 0, 0, None
            Lk = np.asarray(Lk, dtype=np.float64)
            buf = np.ascontiguousarray(Lk.T)   # column-major image of the R matrix
            return ptr(buf, f64p), Lk.shape[0], Lk.shape[1], buf
        lh, lhr, lhc, k1 = img(link_h)
        lw, lwr, lwc, k2 = img(link_w)
        check(self._L.sgl_multi_set_links(self._h, lh, lhr, lhc, lw, lwr, lwc))

    def iterate(self, L1_w, L1_h, L2_w, L2_h):
        t = C.c_double()
        check(self._L.sgl_multi_iterate(self._h, L1_w, L1_h, L2_w, L2_h, C.byref(t)))
        return t.value

    def nmf_run(self, tol, maxit, L1_w, L1_h, L2_w, L2_h, log=None, poll=None):
        n_iter = C.c_int32()
        tr = np.zeros(max(int(maxit), 1))
        cb = make_callbacks(log, poll)
        check(self._L.sgl_multi_nmf_run(self._h, tol, int(maxit), L1_w, L1_h, L2_w, L2_h, C.byref(n_iter), ptr(tr, f64p),
                                        C.byref(cb)))
        return n_iter.value, tr[:n_iter.value].copy()

    def ard_run(self, tol, maxit, L1, L2, seed, inv_density, overfit_threshold, trace_test_mse, log=None, poll=None):
        cap = int(maxit) + 2
        tm, ft, so = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        itv = np.zeros(cap, dtype=np.int32)
        nt, nit = C.c_int32(), C.c_int32()
        cb = make_callbacks(log, poll)
        check(self._L.sgl_multi_ard_run(self._h, tol, int(maxit), L1, L2, int(seed), int(inv_density), overfit_threshold,
                                        int(trace_test_mse), ptr(tm, f64p), ptr(itv, i32p), ptr(ft, f64p), ptr(so, f64p),
                                        C.byref(nt), C.byref(nit), C.byref(cb)))
        q = nt.value
        return dict(test_mse=tm[:q].copy(), iter=itv[:q].copy(), tol=ft[:q].copy(), score_overfit=so[:q].copy(),
                    n_iter=nit.value)

    def get_factors(self):
        nr, nc = self._dims
        W, D, H = np.empty((nr, self.k)), np.empty(self.k), np.empty((nc, self.k))
        check(self._L.sgl_multi_get_factors(self._h, ptr(W, f64p), ptr(D, f64p), ptr(H, f64p)))
        return W, D, H
