"""Independent numpy transcription of singlet's ALS hot path.

TEST INFRASTRUCTURE ONLY.  Written from the reference lines (cited per
function, paths relative to /root/reference), NOT from singlet_oracle.c, so the
two restatements check each other: tests/test_oracle.py asserts bit-for-bit
agreement on small cases, and tests/golden/make_golden.py uses this file to
emit the committed golden vectors.  Column-batched (vectorised ACROSS columns,
strictly sequential WITHIN a column), so every scalar operation happens in the
same order as the reference's per-column loop.  numpy ufuncs never fuse
mul+add, matching an SSE2 build of the reference.

Layout convention: a k x cols column-major matrix (R / Eigen) is a (cols, k)
C-contiguous numpy array.
"""
import numpy as np

U64 = np.uint64
MASK = (1 << 64) - 1


# --- rng: src/singlet.cpp:30-64, 91-95 -------------------------------------
def rand_py(state, i, j):
    """Pure-python-int version (the KAT generator)."""
    i &= MASK
    i ^= (i << 19) & MASK
    i ^= i >> 7
    i ^= (i << 36) & MASK
    x = (state + i) & MASK
    x ^= (x << 38) & MASK
    x ^= x >> 13
    x ^= (x << 23) & MASK
    j &= MASK
    j ^= j >> 7
    j ^= (j << 23) & MASK
    j ^= j >> 8
    x = (x + j) & MASK
    x ^= x >> 7
    x ^= (x << 53) & MASK
    x ^= x >> 4
    return x


def rand_np(state, i, j):
    """Vectorised over uint64 arrays i, j (broadcast)."""
    with np.errstate(over="ignore"):
        i = np.asarray(i, dtype=U64).copy()
        j = np.asarray(j, dtype=U64).copy()
        i ^= i << U64(19)
        i ^= i >> U64(7)
        i ^= i << U64(36)
        x = U64(state) + i
        x ^= x << U64(38)
        x ^= x >> U64(13)
        x ^= x << U64(23)
        j ^= j >> U64(7)
        j ^= j << U64(23)
        j ^= j >> U64(8)
        x = x + j
        x ^= x >> U64(7)
        x ^= x << U64(53)
        x ^= x >> U64(4)
    return x


def draw_np(state, i, j, inv_density):
    return (rand_np(state, i, j) % U64(inv_density)) == 0


# --- helpers -----------------------------------------------------------------
def cor(x, y):
    """src/singlet.cpp:184-197 -- one-pass sums, left to right."""
    x = np.ravel(x)
    y = np.ravel(y)
    n = x.size
    sum_x = sum_y = sum_xy = sum_x2 = sum_y2 = 0.0
    for a, b in zip(x.tolist(), y.tolist()):
        sum_x += a
        sum_y += b
        sum_xy += a * b
        sum_x2 += a * a
        sum_y2 += b * b
    with np.errstate(invalid="ignore", divide="ignore"):
        return float(1 - np.float64(n * sum_xy - sum_x * sum_y) /
                     np.sqrt(np.float64((n * sum_x2 - sum_x * sum_x) * (n * sum_y2 - sum_y * sum_y))))


def aat(F):
    """src/singlet.cpp:200-206.  F: (cols, k).  Left-to-right over columns."""
    cols, k = F.shape
    G = np.zeros((k, k))
    for c in range(cols):
        G += np.outer(F[c], F[c])
    G = np.tril(G) + np.tril(G, -1).T
    G[np.diag_indices(k)] += 1e-15
    return G


def scale(F):
    """src/singlet.cpp:219-225."""
    cols, k = F.shape
    d = np.zeros(k)
    for c in range(cols):
        d += F[c]
    d += 1e-15
    return F / d[None, :], d


def nnls_batch(a, B, X, L1, L2):
    """src/singlet.cpp:229-250 for many columns at once.
    a: (k,k) shared or (ncols,k,k) per column; B, X: (ncols, k), modified in place.
    a is symmetric, so a.col(i) == a[..., i, :]."""
    ncols, k = B.shape
    per_col = a.ndim == 3
    tol = np.ones(ncols)
    it = 0
    sweeps = np.zeros(ncols, dtype=np.int64)
    while it < 100:
        act = (tol / k) > 1e-8
        if not act.any():
            break
        sweeps += act
        tol = np.where(act, 0.0, tol)
        for i in range(k):
            aii = a[:, i, i] if per_col else a[i, i]
            acol = a[:, :, i] if per_col else a[:, i][None, :]
            with np.errstate(divide="ignore", invalid="ignore"):
                diff = B[:, i] / aii
            if L1 != 0:
                diff = diff - L1
            if L2 != 0:
                diff = diff + L2 * X[:, i]
            xi = X[:, i]
            clamp = act & (-diff > xi)
            c2 = clamp & (xi != 0)
            upd = act & ~clamp & (diff != 0)
            if c2.any():
                s = -xi[c2]
                B[c2] = B[c2] - (acol[c2] if per_col else acol) * s[:, None]
                tol[c2] = 1.0
                X[c2, i] = 0.0
            if upd.any():
                X[upd, i] = X[upd, i] + diff[upd]
                B[upd] = B[upd] - (acol[upd] if per_col else acol) * diff[upd][:, None]
                tol[upd] = tol[upd] + np.abs(diff[upd] / (X[upd, i] + 1e-15))
        it += 1
    return sweeps


def _rhs(x, i, p, ncol, F, skip=None):
    """b_c = sum over the column's non-zeros, in stored order, of x * F[row]
    (src/singlet.cpp:341-343); `skip[q]` drops masked entries (:450-457)."""
    k = F.shape[1]
    B = np.zeros((ncol, k))
    cnt = np.diff(p)
    for r in range(int(cnt.max()) if ncol else 0):
        cols = np.nonzero(cnt > r)[0]
        q = p[cols] + r
        if skip is not None:
            keep = ~skip[q]
            cols, q = cols[keep], q[keep]
        B[cols] = B[cols] + x[q][:, None] * F[i[q]]
    return B


def predict(x, i, p, nrow, ncol, F, X, L1, L2):
    """src/singlet.cpp:333-347.  Returns the updated X (copy)."""
    X = X.copy()
    a = aat(F)
    B = _rhs(x, i, p, ncol, F)
    ne = np.nonzero(np.diff(p) > 0)[0]
    Bn, Xn = B[ne], X[ne]
    nnls_batch(a, Bn, Xn, L1, L2)
    X[ne] = Xn
    return X


def predict_mask(x, i, p, nrow, ncol, seed, inv_density, F, X, L1, L2, mask_t):
    """src/singlet.cpp:436-466."""
    X = X.copy()
    k = F.shape[1]
    a = aat(F)
    colidx = np.repeat(np.arange(ncol, dtype=np.uint64), np.diff(p))
    rowidx = i.astype(np.uint64)
    if mask_t:
        skip = draw_np(seed, rowidx, colidx, inv_density)
    else:
        skip = draw_np(seed, colidx, rowidx, inv_density)
    B = _rhs(x, i, p, ncol, F, skip)
    ne = np.nonzero(np.diff(p) > 0)[0]
    a_i = np.empty((ne.size, k, k))
    rows = np.arange(nrow, dtype=np.uint64)
    for t, c in enumerate(ne):
        if mask_t:
            m = draw_np(seed, rows, np.uint64(c), inv_density)
        else:
            m = draw_np(seed, np.uint64(c), rows, inv_density)
        a_i[t] = a - aat(F[m])
    Bn, Xn = B[ne], X[ne]
    nnls_batch(a_i, Bn, Xn, L1, L2)
    X[ne] = Xn
    return X


def mse_test(x, i, p, nrow, ncol, w, d, h, seed, inv_density):
    """src/singlet.cpp:536-568.  w: (m,k), h: (n,k)."""
    w_ = w * d[None, :]
    k = w.shape[1]
    rows = np.arange(nrow, dtype=np.uint64)
    tot = 0.0
    for j in range(ncol):
        m = draw_np(seed, np.uint64(j), rows, inv_density)
        g = np.nonzero(m)[0]
        if g.size == 0:
            tot += 0.0
            continue
        pred = np.zeros(g.size)
        for t in range(k):
            pred = pred + w_[g, t] * h[j, t]
        val = np.zeros(g.size)
        s, e = p[j], p[j + 1]
        pos = np.searchsorted(i[s:e], g)
        ok = pos < (e - s)
        hit = np.zeros(g.size, dtype=bool)
        hit[ok] = i[s:e][pos[ok]] == g[ok]
        val[hit] = x[s:e][pos[hit]]
        sq = (pred - val) ** 2
        ssum = 0.0
        for v in sq.tolist():
            ssum += v
        tot += ssum / g.size
    return tot / ncol


def c_nmf(A, At, tol, maxit, L1_w, L1_h, L2_w, L2_h, w):
    """src/singlet.cpp:638-666.  A, At: objects with x,i,p,nrow,ncol."""
    w = w.copy()
    k = w.shape[1]
    h = np.zeros((A.ncol, k))
    tol_ = 1.0
    it = 0
    tols = []
    d = np.ones(k)
    while it < maxit and tol_ > tol:
        w_it = w.copy()
        h = predict(A.x, A.i, A.p, A.nrow, A.ncol, w, h, L1_h, L2_h)
        h, d = scale(h)
        w = predict(At.x, At.i, At.p, At.nrow, At.ncol, h, w, L1_w, L2_w)
        w, d = scale(w)
        tol_ = cor(w, w_it)
        tols.append(tol_)
        it += 1
    return dict(w=w, d=d, h=h, iter=it, tol=np.array(tols))


def c_project_model(A, w, L1, L2):
    """src/singlet.cpp:405-413.  w in R orientation (m x k or k x m)."""
    w = np.asarray(w, dtype=np.float64)
    if w.shape[0] == A.nrow:
        F = np.ascontiguousarray(w)          # (m, k): already k x m col-major after the transpose
    else:
        F = np.ascontiguousarray(w.T)
    F, d = scale(F)
    h = np.zeros((A.ncol, F.shape[1]))
    h = predict(A.x, A.i, A.p, A.nrow, A.ncol, F, h, L1, L2)
    h, d = scale(h)
    return dict(h=h, d=d)


def c_ard_nmf(A, At, tol, maxit, L1, L2, w, seed, inv_density, overfit_threshold, trace_test_mse):
    """src/singlet.cpp:1090-1152."""
    w = w.copy()
    k = w.shape[1]
    h = np.zeros((A.ncol, k))
    d = np.ones(k)
    tol_ = 1.0
    test_mse, iters, fit_tol, score = [], [], [], []
    it = 0
    while it < maxit and tol_ > tol:
        w_it = w.copy()
        h = predict_mask(A.x, A.i, A.p, A.nrow, A.ncol, seed, inv_density, w, h, L1, L2, False)
        h, d = scale(h)
        w = predict_mask(At.x, At.i, At.p, At.nrow, At.ncol, seed, inv_density, h, w, L1, L2, True)
        w, d = scale(w)
        tol_ = cor(w, w_it)
        if it % trace_test_mse == 0:
            test_mse.append(mse_test(A.x, A.i, A.p, A.nrow, A.ncol, w, d, h, seed, inv_density))
            iters.append(it)
            fit_tol.append(tol_)
            this_err, min_err = test_mse[-1], min(test_mse)
            score.append((this_err - min_err) / (this_err + min_err))
            if score[-1] > overfit_threshold:
                break
        it += 1
    if it % trace_test_mse != 0:
        test_mse.append(mse_test(A.x, A.i, A.p, A.nrow, A.ncol, w, d, h, seed, inv_density))
        iters.append(it)
        fit_tol.append(tol_)
        min_err, this_err = min(test_mse), test_mse[-1]
        score.append((this_err - min_err) / (this_err + min_err))
    return dict(w=w, d=d, h=h, test_mse=np.array(test_mse), iter=np.array(iters, dtype=np.int32),
                tol=np.array(fit_tol), score_overfit=np.array(score), n_iter=it)


def log_normalize(x, p, scale_factor=10000.0):
    """R/PreprocessData.R:34-39 (Seurat::LogNormalize): per column, x / colsum * scale, log1p."""
    out = np.array(x, dtype=np.float64)
    for c in range(len(p) - 1):
        seg = out[p[c]:p[c + 1]]
        s = 0.0
        for v in seg:           # left-to-right, as a scalar loop sums
            s += v
        out[p[c]:p[c + 1]] = np.log1p(seg / s * scale_factor)
    return out


def weight_by_split(x, p, split_by, n_groups):
    """src/singlet.cpp:119-144."""
    out = np.array(x, dtype=np.float64)
    sums = np.zeros(n_groups)
    for j in range(len(p) - 1):
        for v in out[p[j]:p[j + 1]]:
            sums[split_by[j]] += v
    sums[1:] /= sums[0]
    for i in range(len(p) - 1):
        if split_by[i] != 0:
            out[p[i]:p[i + 1]] /= sums[split_by[i]]
    return out
