"""The oracle pinned on the CPU: hash KATs, C restatement == independent numpy transcription
(bit for bit), both == the committed golden vectors, generator statistics, pbmc3k fixture."""
import os

import numpy as np
import pytest

from oracle import np_transcription as npt

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# derived by literal transcription of src/singlet.cpp:30-64 with 64-bit wrap
KATS = [((123, 0, 0), 0x692656729eb6707c), ((123, 1, 2), 0xd1c4f6746e22623f), ((123, 2, 1), 0xa171855b28d239a7),
        ((123, 999999, 29999), 0x3ffec3e4f2d21eea), ((2147483647, 5, 7), 0x3b72606ab1a2d601),
        ((1, 0, 1), 0x000112648b36e912)]


def _csc(ora, g):
    m, n = int(g["dim"][0]), int(g["dim"][1])
    return ora.CSC(g["Ax"], g["Ai"], g["Ap"], m, n), ora.CSC(g["Atx"], g["Ati"], g["Atp"], n, m)


def test_hash_kats(ora):
    for (s, i, j), v in KATS:
        assert ora.rng_rand(s, i, j) == v
        assert npt.rand_py(s, i, j) == v
        assert int(npt.rand_np(s, np.uint64(i), np.uint64(j))) == v
    assert [v % 20 for _, v in KATS] == [4, 7, 7, 6, 9, 18]
    g = np.load(os.path.join(GOLD, "rng_kat.npz"))
    for (s, i, j), v in zip(g["args"].tolist(), g["out"].tolist()):
        assert ora.rng_rand(s, i, j) == v


def test_draw_rate_and_mask_agree(ora):
    m = ora.rng_mask(123, 0, 500, 2000, 20)
    assert abs(m.mean() - 0.05) < 0.002
    cells = np.arange(500, dtype=np.uint64)[:, None]
    genes = np.arange(2000, dtype=np.uint64)[None, :]
    assert np.array_equal(m.astype(bool), npt.draw_np(123, cells, genes, 20))


def test_generator_density_and_transpose(ora):
    A = ora.synth_csc(700, 900, 20)
    assert abs(A.nnz / (700 * 900) - 0.05) < 0.002
    assert A.x.min() > 0 and len(np.unique(A.x)) == 16
    for c in range(0, 900, 97):                       # rows ascending inside every column
        r = A.i[A.p[c]:A.p[c + 1]]
        assert np.all(np.diff(r) > 0)
    At = A.t()
    assert np.array_equal(At.to_dense(), A.to_dense().T)
    w = ora.synth_winit(5, 700)
    assert w.min() > 0 and w.max() < 1 and abs(w.mean() - 0.5) < 0.02


@pytest.mark.parametrize("L1,L2", [(0.0, 0.0), (0.01, 0.0), (0.01, 0.02)])
def test_c_oracle_equals_numpy_transcription(ora, L1, L2):
    A = ora.synth_csc(150, 190, 12)
    At = A.t()
    w0 = ora.synth_winit(6, 150)
    a = ora.c_nmf(A, At, 0.0, 4, L1, L1, L2, L2, 0, w0)
    b = npt.c_nmf(A, At, 0.0, 4, L1, L1, L2, L2, w0)
    for key in ("w", "h", "d", "tol"):
        assert np.array_equal(a[key], b[key]), key


def test_c_oracle_masked_path_equals_numpy_transcription(ora):
    A = ora.synth_csc(110, 140, 10)
    At = A.t()
    w0 = ora.synth_winit(5, 110)
    a = ora.c_ard_nmf(A, At, 0.0, 5, 0.01, 0.0, 0, w0, 31, 10, 1e-3, 2)
    b = npt.c_ard_nmf(A, At, 0.0, 5, 0.01, 0.0, w0, 31, 10, 1e-3, 2)
    for key in ("w", "h", "d", "test_mse", "iter", "tol", "score_overfit"):
        assert np.array_equal(a[key], b[key]), key


@pytest.mark.parametrize("name", ["nmf_k8_l1_0", "nmf_k8_l1_01", "nmf_k8_l1_01_l2_01", "nmf_k30"])
def test_golden_c_nmf(ora, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    A, At = _csc(ora, g)
    L1, L2 = float(g["L1"]), float(g["L2"])
    r = ora.c_nmf(A, At, 0.0, int(g["maxit"]), L1, L1, L2, L2, 0, g["w0"])
    for key in ("w", "h", "d", "tol"):
        assert np.array_equal(r[key], g[key]), key


def test_golden_inputs_come_from_the_generator(ora):
    g = np.load(os.path.join(GOLD, "nmf_k8_l1_0.npz"))
    A = ora.synth_csc(300, 400, 20)
    assert np.array_equal(A.x, g["Ax"]) and np.array_equal(A.i, g["Ai"]) and np.array_equal(A.p, g["Ap"])
    assert np.array_equal(ora.synth_winit(8, 300), g["w0"])


def test_golden_ard_and_project(ora):
    g = np.load(os.path.join(GOLD, "ard_k6.npz"))
    A, At = _csc(ora, g)
    r = ora.c_ard_nmf(A, At, 0.0, int(g["maxit"]), float(g["L1"]), float(g["L2"]), 0, g["w0"], int(g["seed"]),
                      int(g["inv_density"]), float(g["overfit_threshold"]), int(g["trace_test_mse"]))
    for key in ("w", "h", "d", "test_mse", "iter", "tol", "score_overfit"):
        assert np.array_equal(r[key], g[key]), key
    g = np.load(os.path.join(GOLD, "project_k5.npz"))
    A, _ = _csc(ora, g)
    r = ora.c_project_model(A, g["w"], float(g["L1"]), float(g["L2"]))
    assert np.array_equal(r["h"], g["h"]) and np.array_equal(r["d"], g["d"])
    r2 = ora.c_project_model(A, np.ascontiguousarray(g["w"].T), float(g["L1"]), float(g["L2"]))
    assert np.array_equal(r2["h"], g["h"])


def test_rcpp_predict_against_transcription(ora):
    """Rcpp_predict (src/singlet.cpp:350-367): transposition rule and no scaling."""
    from oracle import np_transcription as npt
    A = ora.synth_csc(60, 90, 20)
    rng = np.random.default_rng(3)
    for shape in ((60, 7), (7, 60), (60, 60)):
        w = rng.random(shape)
        F = w if (shape[0] == 60 and shape[1] != 60) else np.ascontiguousarray(w.T)
        ref = npt.predict(A.x, A.i, A.p, A.nrow, A.ncol, np.ascontiguousarray(F), np.zeros((90, F.shape[1])), 0.01, 0.0)
        assert np.array_equal(ora.rcpp_predict(A, w, 0.01, 0.0), ref), shape


def test_staging_ops_against_transcription(ora):
    """LogNormalize (R/PreprocessData.R:34-39) and weight_by_split (src/singlet.cpp:119-144)."""
    from oracle import np_transcription as npt
    A = ora.synth_csc(120, 90, 8)
    counts = ora.CSC(np.round(np.expm1(A.x)), A.i, A.p, A.nrow, A.ncol)   # integer "counts"
    ln = ora.log_normalize(counts, 1e4)
    ref = npt.log_normalize(counts.x, counts.p, 1e4)
    assert np.abs(ln.x - ref).max() <= 4e-16 * np.abs(ref).max()   # libm vs numpy log1p: last-bit differences
    assert np.array_equal(ln.i, counts.i) and np.array_equal(ln.p, counts.p)
    sb = np.random.default_rng(0).integers(0, 3, A.ncol)
    ws = ora.weight_by_split(ln, sb, 3)
    assert np.array_equal(ws.x, npt.weight_by_split(ln.x, ln.p, sb, 3))
    per_cell = np.repeat(sb, np.diff(A.p))
    totals = [ws.x[per_cell == g].sum() for g in range(3)]
    assert np.allclose(totals, totals[0], rtol=1e-12)               # every group now weighs as much as group 0
    assert np.array_equal(ws.x[per_cell == 0], ln.x[per_cell == 0])  # group 0 untouched (l.137)


def test_c_linked_nmf_oracle(ora):
    """No links == c_nmf bit for bit; a zero link pins its coefficient at zero; mismatched links are ignored."""
    A = ora.synth_csc(80, 60, 8)
    At = A.t()
    w0 = ora.synth_winit(5, 80)
    r0 = ora.c_nmf(A, At, 0.0, 3, 0.01, 0.01, 0.0, 0.0, 0, w0)
    r1 = ora.c_linked_nmf(A, At, 0.0, 3, 0.01, 0.0, 0, w0, None, np.ones((1, 1)))
    assert np.array_equal(r0["w"], r1["w"]) and np.array_equal(r0["h"], r1["h"]) and np.array_equal(r0["d"], r1["d"])
    lh = (np.random.default_rng(0).random((5, 60)) < 0.7).astype(float)
    r2 = ora.c_linked_nmf(A, At, 0.0, 3, 0.01, 0.0, 0, w0, lh, None)
    assert np.all(r2["h"][lh.T == 0] == 0) and not np.array_equal(r2["h"], r0["h"])


def test_c_nmf_dense_oracle(ora):
    """Dense front-end: equals the sparse loop bit for bit when no column is empty; an all-zero column is
    still solved (src/singlet.cpp:370-381 has no skip), which the sparse loop would leave untouched."""
    A = ora.synth_csc(60, 50, 6)
    D = np.zeros((60, 50))
    for c in range(50):
        D[A.i[A.p[c]:A.p[c + 1]], c] = A.x[A.p[c]:A.p[c + 1]]
    w0 = ora.synth_winit(4, 60)
    r0 = ora.c_nmf(A, A.t(), 0.0, 3, 0.01, 0.01, 0.0, 0.0, 0, w0)
    r1 = ora.c_nmf_dense(D, 0.0, 3, 0.01, 0.01, 0.0, 0.0, 0, w0)
    assert np.array_equal(r0["w"], r1["w"]) and np.array_equal(r0["h"], r1["h"])
    D[7, :] = 0.0     # gene 7 never expressed: its w column is clamped by the L1 step instead of kept
    r2 = ora.c_nmf_dense(D, 0.0, 3, 0.01, 0.01, 0.0, 0.0, 0, w0)
    assert np.all(r2["w"][7] == 0.0)


def test_nnls_quirks(ora):
    """SURVEY.md 8a quirks 2-5 on hand-made cases."""
    G = np.array([[2.0, 0.5], [0.5, 1.0]])
    # (4) a negative step on x == 0 does nothing; (5) first sweep always runs
    x, b, it = ora.nnls(G, np.array([-1.0, -1.0]), np.zeros(2))
    assert np.all(x == 0) and it == 1
    # (2) L1 is subtracted from every step of every sweep
    x0, _, _ = ora.nnls(G, np.array([1.0, 1.0]), np.zeros(2), 0.0, 0.0)
    x1, _, _ = ora.nnls(G, np.array([1.0, 1.0]), np.zeros(2), 0.01, 0.0)
    assert np.all(x1 < x0)
    # (3) a clamp overwrites the sweep's tol with 1, forcing another sweep
    x, _, it = ora.nnls(G, np.array([-5.0, 3.0]), np.array([1.0, 0.0]))
    assert x[0] == 0 and it >= 2
    # cap of 100 sweeps (uint8 counter, src/singlet.cpp:231)
    Gb = np.array([[1.0, 0.999999], [0.999999, 1.0]])
    _, _, it = ora.nnls(Gb, np.array([1.0, 1.0000001]), np.zeros(2))
    assert it == 100


def test_pbmc3k_fixture_and_config1(ora):
    """BASELINE config 1: pbmc3k 13714 x 2700, LogNormalize, k = 10, on the CPU path (plumbing)."""
    g = np.load(os.path.join(GOLD, "pbmc3k_counts.npz"))
    p, dim = g["p"], g["dim"]
    i = g["di"].astype(np.int64)
    for c in range(dim[1]):                      # undo the per-column delta coding
        s, e = p[c], p[c + 1]
        i[s:e] = np.cumsum(i[s:e])
    assert tuple(dim) == (13714, 2700) and p[-1] == 2282976
    per_cell = np.diff(p)
    assert (per_cell.min(), int(np.median(per_cell)), per_cell.max()) == (212, 816, 3400)   # SURVEY.md 8 [probe]
    x = g["x"].astype(np.float64)
    colsum = np.add.reduceat(x, p[:-1])
    xn = np.log1p(x / np.repeat(colsum, per_cell) * 1e4)   # Seurat::LogNormalize (R/PreprocessData.R:35)
    A = ora.CSC(xn, i.astype(np.int32), p, dim[0], dim[1])
    w0 = ora.synth_winit(10, dim[0])
    r = ora.c_nmf(A, A.t(), 1e-5, 3, 0.01, 0.01, 0.0, 0.0, 0, w0)
    assert r["iter"] == 3 and np.all(np.isfinite(r["w"])) and np.all(r["h"] >= 0)
    assert np.all(np.diff(r["tol"]) < 0)         # converging
    assert abs(r["w"].sum(axis=0) - 1).max() < 1e-9   # rows of W sum to 1 after scale()


def test_list_and_dense_ard_restatements_agree_with_the_plain_ones(ora):
    """The chunk-list loops (src/singlet.cpp:384-402, 469-503, 571-607: running column offset) and the dense
    masked loops (:506-533, 608-632) restated in the oracle: on a matrix without empty columns they are the
    plain c_nmf / c_ard_nmf bit for bit, however the columns are cut."""
    A = ora.synth_csc(60, 90, 5)
    At = A.t()
    w0 = ora.synth_winit(5, 60)

    def split(M, cuts):
        b = [0] + cuts + [M.ncol]
        return [ora.CSC(M.x[M.p[lo]:M.p[hi]], M.i[M.p[lo]:M.p[hi]], M.p[lo:hi + 1] - M.p[lo], M.nrow, hi - lo)
                for lo, hi in zip(b[:-1], b[1:])]
    r1 = ora.c_nmf(A, At, 0.0, 4, 0.01, 0.01, 0, 0, 0, w0)
    r2 = ora.c_nmf_sparse_list(split(A, [20, 55]), split(At, [7, 30, 31]), 0.0, 4, 0.01, 0.0, 0, w0)
    assert np.array_equal(r1["w"], r2["w"]) and np.array_equal(r1["h"], r2["h"]) and np.array_equal(r1["tol"], r2["tol"])
    a1 = ora.c_ard_nmf(A, At, 0.0, 5, 0.01, 0.0, 0, w0, 7, 10, 1e9, 2)
    a2 = ora.c_ard_nmf_sparse_list(split(A, [1, 89]), split(At, [59]), 0.0, 5, 0.01, 0.0, 0, w0, 7, 10, 1e9, 2)
    a3 = ora.c_ard_nmf_dense(A.to_dense(), 0.0, 5, 0.01, 0.0, 0, w0, 7, 10, 1e9, 2)
    for a in (a2, a3):
        assert list(a["iter"]) == list(a1["iter"]) == [0, 2, 4, 5]
        assert np.array_equal(a["w"], a1["w"]) and np.array_equal(a["h"], a1["h"]) and np.array_equal(a["test_mse"], a1["test_mse"])
    # an all-zero gene: skipped by the sparse loops (stale w column), solved by the dense one
    D = A.to_dense()
    D[3, :] = 0.0
    import scipy.sparse as sp
    S = sp.csc_matrix(D)
    As = ora.CSC(S.data, S.indices, S.indptr, 60, 90)
    s1 = ora.c_ard_nmf(As, As.t(), 0.0, 3, 0.01, 0.0, 0, w0, 7, 10, 1e9, 1)
    s3 = ora.c_ard_nmf_dense(D, 0.0, 3, 0.01, 0.0, 0, w0, 7, 10, 1e9, 1)
    assert not np.array_equal(s1["w"][3], s3["w"][3])


# ---- the integer part pinned to the REFERENCE's own code (oracle/make_ref.sh, tests/golden/make_rng_ref.py) ----
def _rng_ref():
    g = np.load(os.path.join(GOLD, "rng_ref.npz"))
    grids = np.unpackbits(g["draw"], axis=-1)[..., :int(g["draw_shape"][-1])]
    return g, grids


def test_hash_equals_the_reference_rng_class(ora):
    """101 152 (state, i, j) -> rand(i, j) triples computed by the reference's `rng` class itself
    (src/singlet.cpp:6-114 compiled as it lies in the reference tree): C oracle and numpy transcription bit-exact."""
    g, _ = _rng_ref()
    state, i, j, exp = g["state"], g["i"], g["j"], g["rand2"]
    assert i.size >= 100000
    for s in np.unique(state):
        sel = state == s
        assert np.array_equal(npt.rand_np(int(s), i[sel], j[sel]), exp[sel])
    step = 7   # the C oracle one call per triple: every 7th, plus all edge values (the first 144 of every state block)
    idx = np.unique(np.concatenate([np.arange(0, i.size, step)] + [b + np.arange(144) for b in np.nonzero(np.diff(state, prepend=state[0] + 1))[0]]))
    got = np.array([ora.rng_rand(int(state[q]), int(i[q]), int(j[q])) for q in idx], dtype=np.uint64)
    assert np.array_equal(got, exp[idx])
    for (s, a, b), v in KATS:   # the hand-derived known answers are the reference's too
        hit = np.nonzero((state == s) & (i == a) & (j == b))[0]
        assert hit.size == 0 or int(exp[hit[0]]) == v


def test_mask_draws_equal_the_reference_rng_class(ora):
    """draw(cell, gene, inv_density) grids (24 cells x 1500 genes at cell offsets 0 and 999 000, seven densities)
    from the reference's class: the oracle's mask (what every masked parity test compares with) is identical."""
    g, grids = _rng_ref()
    seed = int(g["draw_state"])
    nc, ng = grids.shape[2], grids.shape[3]
    for a, inv in enumerate(g["draw_inv_density"].tolist()):
        for b, c0 in enumerate(g["draw_cell0"].tolist()):
            assert np.array_equal(ora.rng_mask(seed, c0, nc, ng, inv), grids[a, b]), (inv, c0)
            cells = (np.uint64(c0) + np.arange(nc, dtype=np.uint64))[:, None]
            assert np.array_equal(npt.draw_np(seed, cells, np.arange(ng, dtype=np.uint64)[None, :], inv), grids[a, b].astype(bool))


def test_live_reference_build_when_present(ora):
    """oracle/_ref/librng_ref.so (built by oracle/make_ref.sh from the reference tree; travels to the GPU box as a
    build product) against the oracle on fresh random inputs -- skipped only where it was never built."""
    import ctypes as C
    path = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "librng_ref.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    L = C.CDLL(os.path.abspath(path))
    u64p = C.POINTER(C.c_uint64)
    L.ref_rand2.argtypes = [C.c_uint64, u64p, u64p, C.c_int64, u64p]
    L.ref_draw_grid.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.c_uint64, C.c_int64, C.POINTER(C.c_uint8)]
    r = np.random.default_rng(11)
    i = r.integers(0, 2 ** 64, 3000, dtype=np.uint64, endpoint=False)
    j = r.integers(0, 2 ** 64, 3000, dtype=np.uint64, endpoint=False)
    out = np.empty_like(i)
    for s in (0, 77, 2 ** 64 - 1):
        L.ref_rand2(s, i.ctypes.data_as(u64p), j.ctypes.data_as(u64p), i.size, out.ctypes.data_as(u64p))
        assert np.array_equal(out, npt.rand_np(s, i, j))
        assert all(ora.rng_rand(s, int(i[q]), int(j[q])) == int(out[q]) for q in range(0, 3000, 10))
    buf = np.empty((50, 700), dtype=np.uint8)
    for inv in (4, 20, 999):
        L.ref_draw_grid(9, inv, 123456, 50, 0, 700, buf.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert np.array_equal(buf, ora.rng_mask(9, 123456, 50, 700, inv))


def test_oracle_nnls_solves_the_problem_scipy_solves(ora):
    """An independent check of WHAT the restated nnls (src/singlet.cpp:229-250) computes, by a different algorithm: for a positive
    definite Gram a = F F^T and right-hand side b, coordinate descent on 1/2 x^T a x - b^T x over x >= 0 has the minimiser that
    scipy's active-set NNLS finds for min |F^T x - y| with F y = b.  The reference's loop stops on a relative-change test
    (tol / k <= 1e-8) or after 100 sweeps, so on a well-conditioned Gram the two agree to ~1e-7 of the largest entry (on positive, correlated factors only to ~1e-3), with the same support up to entries at that level, and the KKT
    conditions hold at the oracle's solution.  Not a pin of the floating-point bits (nothing in this image can be: DESIGN.md
    "Oracle status") -- a check that the restatement solves the right problem."""
    from scipy.optimize import nnls as scipy_nnls
    rng = np.random.default_rng(11)
    for k, rows in ((6, 200), (20, 700), (50, 2000)):     # rows >> k: a well-conditioned Gram, so that the sweep cap and the loose stop do not bind
        for _ in range(5):
            F = rng.normal(size=(k, rows))     # k x rows, independent entries: a = F F^T is well conditioned (rows >> k)
            a = F @ F.T
            y = rng.normal(size=rows) + 0.5
            b = F @ y
            x, _, it = ora.nnls(a, b.copy(), np.zeros(k))
            xs, _ = scipy_nnls(F.T, y, maxiter=50 * k)
            assert 0 < it <= 100
            scale = max(np.abs(xs).max(), 1e-12)
            assert np.abs(x - xs).max() <= 1e-6 * scale, (k, np.abs(x - xs).max() / scale)
            assert np.array_equal(x > 1e-5 * scale, xs > 1e-5 * scale)
            # KKT at the oracle's x: gradient g = a x - b is ~0 on the support and >= 0 off it
            g = a @ x - b
            on = x > 1e-6 * scale
            gs = np.abs(b).max()
            assert np.abs(g[on]).max(initial=0.0) <= 1e-6 * gs and g[~on].min(initial=0.0) >= -1e-6 * gs
