"""End-to-end parity of the ALS loops against the CPU oracle through the
reference-shaped entry points (c_nmf / c_ard_nmf / c_project_model)."""
import numpy as np
import pytest

from conftest import rel_fro, same_zero_pattern, to_dgc

pytestmark = pytest.mark.gpu
TOL = 1e-9   # asserted; the north star allows 1e-5


def _check(got, ref, keys=("w", "h", "d")):
    for key in keys:
        g = got[key].T if got[key].ndim == 2 else got[key]
        assert rel_fro(g, ref[key]) < TOL, key
        if g.ndim == 2:
            assert same_zero_pattern(g, ref[key]), key


@pytest.mark.parametrize("m,n,k,L1,L2,maxit", [
    (300, 400, 8, 0.0, 0.0, 5), (300, 400, 8, 0.01, 0.0, 5), (300, 400, 8, 0.01, 0.01, 5),
    (500, 260, 30, 0.01, 0.0, 4), (257, 1031, 50, 0.01, 0.0, 3), (200, 300, 1, 0.0, 0.0, 3),
    (150, 200, 64, 0.01, 0.0, 2), (150, 220, 70, 0.01, 0.0, 2),
    # ranks above 64: two-part tiled accumulate, split Gram, lane NNLS with x in scratch (100), one wave per SIMD
    # (120, 128), and above 128 the plain CSC accumulate + wave-per-column NNLS (130, 200)
    (260, 330, 100, 0.01, 0.0, 2), (250, 300, 120, 0.01, 0.0, 2), (270, 310, 128, 0.01, 0.01, 2),
    (280, 300, 130, 0.01, 0.0, 2), (300, 420, 200, 0.01, 0.0, 2),
    # above 256: the any-rank instances (16 coordinates per lane in the wave NNLS, Gram in several launches)
    (400, 520, 300, 0.01, 0.0, 2), (700, 640, 600, 0.0, 0.0, 1)])
def test_c_nmf_parity(sa, ora, m, n, k, L1, L2, maxit):
    A = ora.synth_csc(m, n, 20)
    At = A.t()
    w0 = ora.synth_winit(k, m)
    ref = ora.c_nmf(A, At, 0.0, maxit, L1, L1, L2, L2, 0, w0)
    got = sa.c_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, maxit, False, L1, L1, L2, L2, 0, w0.T)
    _check(got, ref)
    assert got["iter"] == ref["iter"] == maxit
    assert np.allclose(got["tol"], ref["tol"], rtol=1e-8, atol=0)


def test_c_nmf_without_At_and_distinct_penalties(sa, ora):
    A = ora.synth_csc(280, 350, 10)
    w0 = ora.synth_winit(9, 280)
    ref = ora.c_nmf(A, A.t(), 0.0, 4, 0.02, 0.005, 0.01, 0.0, 0, w0)
    got = sa.c_nmf(to_dgc(sa, A), None, 0.0, 4, False, 0.02, 0.005, 0.01, 0.0, 0, w0.T)
    _check(got, ref)


def test_c_nmf_stop_decision(sa, ora):
    """tol > 0: the loop must stop at the same iteration as the oracle (well separated case)."""
    A = ora.synth_csc(300, 400, 20)
    w0 = ora.synth_winit(6, 300)
    ref = ora.c_nmf(A, A.t(), 1e-2, 100, 0.01, 0.01, 0, 0, 0, w0)
    got = sa.c_nmf(to_dgc(sa, A), None, 1e-2, 100, False, 0.01, 0.01, 0, 0, 0, w0.T)
    assert got["iter"] == ref["iter"] and 1 < ref["iter"] < 100
    _check(got, ref)


@pytest.mark.parametrize("k", [5, 30, 70, 104])
def test_c_nmf_empty_columns_keep_stale_values(sa, ora, k):
    """(k = 5 / 30: the lane solve resp. four columns per wave; 70 / 104: the generated two-lane solve without / with x in the
    accumulator registers -- a lane pair without a column must stay out of the loads, the sweeps and the stores)"""
    rng = np.random.default_rng(5)
    D = (rng.random((120, 150)) < 0.1) * (rng.random((120, 150)) + 0.5)
    D[:, 7] = 0      # empty cell: h[:, 7] stays 0 then is only rescaled
    D[33, :] = 0     # empty gene: w[:, 33] keeps its (rescaled) initial values
    from test_gpu_ops import _csc_from_dense
    A = ora.CSC(*_csc_from_dense(D))
    w0 = ora.synth_winit(k, 120)
    ref = ora.c_nmf(A, A.t(), 0.0, 3, 0.01, 0.01, 0, 0, 0, w0)
    got = sa.c_nmf(to_dgc(sa, A), None, 0.0, 3, False, 0.01, 0.01, 0, 0, 0, w0.T)
    _check(got, ref)
    assert np.all(got["h"][:, 7] == 0) and np.all(got["w"][:, 33] > 0)


@pytest.mark.parametrize("k", [11, 90, 140])
@pytest.mark.parametrize("orient", ["m_by_k", "k_by_m"])
def test_c_project_model(sa, ora, orient, k):
    A = ora.synth_csc(300, 410, 20)
    w = np.random.default_rng(1).random((300, k))
    win = w if orient == "m_by_k" else w.T.copy()
    ref = ora.c_project_model(A, win, 0.01, 0.0)
    got = sa.c_project_model(to_dgc(sa, A), win, 0.01, 0.0, 0)
    _check(got, ref, ("h", "d"))


@pytest.mark.parametrize("shape", [(300, 11), (11, 300), (40, 40)])
def test_rcpp_predict(sa, ora, shape):
    """Rcpp_predict transposes w only if w.rows() == A.rows() and w.cols() != A.rows()
    (src/singlet.cpp:351): the square case is NOT transposed, unlike c_project_model."""
    A = ora.synth_csc(max(shape), 410, 20)
    w = np.random.default_rng(2).random(shape)
    ref = ora.rcpp_predict(A, w, 0.01, 0.0)
    got = sa.Rcpp_predict(to_dgc(sa, A), w, 0.01, 0.0, 0)
    assert rel_fro(got.T, ref) < TOL and same_zero_pattern(got.T, ref)


@pytest.mark.parametrize("k,trace,maxit", [(6, 1, 4), (8, 2, 5), (5, 3, 4), (17, 2, 3), (20, 1, 3), (30, 2, 3), (36, 2, 2), (49, 2, 2), (50, 2, 2), (52, 2, 2), (66, 2, 2), (70, 2, 2), (83, 2, 2), (90, 1, 2), (100, 2, 2), (116, 2, 2),
                                            # 4 < k % 16 <= 8 at NT = 3 .. 5: remainder rows as two quads of quarter-MFMAs (mask_gram_list_kernel<NT, 1, 0, 8>)
                                            (53, 2, 2), (55, 2, 2), (56, 2, 2), (69, 2, 2), (72, 2, 2), (87, 2, 2), (88, 2, 2),
                                            # above 128: Gram downdates on the VALU in pair ranges, wave NNLS (any-rank path)
                                            (140, 2, 2), (260, 1, 1)])
def test_c_ard_nmf_parity(sa, ora, k, trace, maxit):
    m, n = (220, 260) if k <= 100 else (900, 1000)   # enough data for every factor to stay alive
    A = ora.synth_csc(m, n, 20)
    At = A.t()
    w0 = ora.synth_winit(k, m)
    ref = ora.c_ard_nmf(A, At, 0.0, maxit, 0.01, 0.0, 0, w0, 77, 20, 1e-3, trace)
    got = sa.c_ard_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, maxit, False, 0.01, 0.0, 0, w0.T, 77, 20, 1e-3, trace)
    _check(got, ref)
    assert np.array_equal(got["iter"], ref["iter"])
    assert np.allclose(got["test_mse"], ref["test_mse"], rtol=1e-9, atol=0)
    assert np.allclose(got["tol"], ref["tol"], rtol=1e-7, atol=0)
    assert np.allclose(got["score_overfit"], ref["score_overfit"], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("k", [113, 116, 128])
def test_c_ard_nmf_parity_quad_solve_above_112(sa, ora, k, monkeypatch):
    """Ranks 113 - 128 take the four-columns-per-wave solve on global Grams only for launches of 65 536 columns or
    more (shorter ones keep the wave kernel): force it for a small problem and hold the fit against the oracle."""
    monkeypatch.setenv("SGL_NNLS_QUAD_GLOBAL_MIN_COLS", "1")
    m, n = 900, 1000
    A = ora.synth_csc(m, n, 20)
    At = A.t()
    w0 = ora.synth_winit(k, m)
    ref = ora.c_ard_nmf(A, At, 0.0, 2, 0.01, 0.0, 0, w0, 77, 20, 1e-3, 2)
    got = sa.c_ard_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, 2, False, 0.01, 0.0, 0, w0.T, 77, 20, 1e-3, 2)
    _check(got, ref)
    assert np.allclose(got["test_mse"], ref["test_mse"], rtol=1e-9, atol=0)


@pytest.mark.parametrize("k", [3, 16, 17, 31, 40, 44, 47, 48])
def test_c_ard_nmf_global_quad_solve_equals_the_lds_one(sa, ora, k, monkeypatch):
    """The masked solve takes four columns per wave on the column's Gram in global memory from k = 40 on long launches (47 on
    short ones) and on its LDS triangle below: same arithmetic in the same order, so the fits are the same bits whichever
    runs (SGL_NNLS_QUAD_GLOBAL_FROM moves the limit), and the oracle's to 1e-9."""
    m, n = 220, 260
    A = ora.synth_csc(m, n, 20)
    At = A.t()
    w0 = ora.synth_winit(k, m)
    ref = ora.c_ard_nmf(A, At, 0.0, 3, 0.01, 0.0, 0, w0, 77, 20, 1e-3, 1)
    got = {}
    for name, frm in (("lds", "1000"), ("global", "1")):
        monkeypatch.setenv("SGL_NNLS_QUAD_GLOBAL_FROM", frm)
        got[name] = sa.c_ard_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, 3, False, 0.01, 0.0, 0, w0.T, 77, 20, 1e-3, 1)
    for key in ("w", "h", "d", "test_mse", "tol"):
        assert np.array_equal(got["lds"][key], got["global"][key]), key
    _check(got["global"], ref)
    assert np.allclose(got["global"]["test_mse"], ref["test_mse"], rtol=1e-9, atol=0)


def test_c_ard_nmf_overfit_break(sa, ora):
    """A tiny overfit threshold makes the reference break out of the loop early; same here."""
    A = ora.synth_csc(200, 240, 20)
    w0 = ora.synth_winit(12, 200)
    ref = ora.c_ard_nmf(A, A.t(), 0.0, 30, 0.0, 0.0, 0, w0, 5, 10, 1e-7, 1)
    got = sa.c_ard_nmf(to_dgc(sa, A), None, 0.0, 30, False, 0.0, 0.0, 0, w0.T, 5, 10, 1e-7, 1)
    assert np.array_equal(got["iter"], ref["iter"])
    assert len(ref["iter"]) < 30
    _check(got, ref)


def test_step_api_matches_run(sa, ora, ctx):
    """The step-level operators chained by the host equal sgl_nmf_run (what a sharded host does)."""
    A = ora.synth_csc(260, 300, 20)
    w0 = ora.synth_winit(10, 260)
    ref = ora.c_nmf(A, A.t(), 0.0, 3, 0.01, 0.01, 0, 0, 0, w0)
    ctx.upload(to_dgc(sa, A), None)
    ctx.fit_init(10, w0)
    tols = []
    for _ in range(3):
        ctx.step_begin()
        ctx.step_h(0.01, 0.0)
        ctx.step_scale_h()
        ctx.step_w(0.01, 0.0)
        tols.append(ctx.step_scale_w())
    W, d, H = ctx.get_factors()
    assert rel_fro(W, ref["w"]) < TOL and rel_fro(H, ref["h"]) < TOL and rel_fro(d, ref["d"]) < TOL
    assert np.allclose(tols, ref["tol"], rtol=1e-8)


# ---- committed golden vectors (tests/golden/, generated by make_golden.py) ------------------
import os  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["nmf_k8_l1_0", "nmf_k8_l1_01", "nmf_k8_l1_01_l2_01", "nmf_k30"])
def test_golden_c_nmf_hip(sa, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    m, n = int(g["dim"][0]), int(g["dim"][1])
    A = sa.dgCMatrix(g["Ax"], g["Ai"], g["Ap"], (m, n))
    At = sa.dgCMatrix(g["Atx"], g["Ati"], g["Atp"], (n, m))
    L1, L2 = float(g["L1"]), float(g["L2"])
    got = sa.c_nmf(A, At, 0.0, int(g["maxit"]), False, L1, L1, L2, L2, 0, g["w0"].T)
    _check(got, g)
    assert np.allclose(got["tol"], g["tol"], rtol=1e-8, atol=0)


def test_golden_ard_and_project_hip(sa):
    g = np.load(os.path.join(GOLD, "ard_k6.npz"))
    m, n = int(g["dim"][0]), int(g["dim"][1])
    A = sa.dgCMatrix(g["Ax"], g["Ai"], g["Ap"], (m, n))
    got = sa.c_ard_nmf(A, None, 0.0, int(g["maxit"]), False, float(g["L1"]), float(g["L2"]), 0, g["w0"].T,
                       int(g["seed"]), int(g["inv_density"]), float(g["overfit_threshold"]), int(g["trace_test_mse"]))
    _check(got, g)
    assert np.array_equal(got["iter"], g["iter"])
    assert np.allclose(got["test_mse"], g["test_mse"], rtol=1e-9, atol=0)
    g = np.load(os.path.join(GOLD, "project_k5.npz"))
    A = sa.dgCMatrix(g["Ax"], g["Ai"], g["Ap"], (m, n))
    got = sa.c_project_model(A, g["w"], float(g["L1"]), float(g["L2"]), 0)
    _check(got, g, ("h", "d"))


def test_pbmc3k_config1(sa, ora):
    """BASELINE config 1: pbmc3k 13714 x 2700 (the reference's bundled data), LogNormalize, k = 10,
    L1 = 0.01, through run_nmf's entry point; HIP == oracle at equal iteration count, and the
    RunNMF-level post-processing (factors sorted by d) applied on top."""
    g = np.load(os.path.join(GOLD, "pbmc3k_counts.npz"))
    p, dim = g["p"], g["dim"]
    i = g["di"].astype(np.int64)
    for c in range(dim[1]):
        s, e = p[c], p[c + 1]
        i[s:e] = np.cumsum(i[s:e])
    x = g["x"].astype(np.float64)
    counts = ora.CSC(x, i.astype(np.int32), p, dim[0], dim[1])
    A = ora.log_normalize(counts, 1e4)                       # oracle: LogNormalize then the fit
    w0 = ora.synth_winit(10, dim[0])
    ref = ora.c_nmf(A, A.t(), 0.0, 6, 0.01, 0.01, 0.0, 0.0, 0, w0)
    dA = sa.PreprocessData(to_dgc(sa, counts))               # device: PreprocessData.dgCMatrix then c_nmf
    assert rel_fro(dA.x, A.x) < 1e-14
    got = sa.c_nmf(dA, None, 0.0, 6, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    _check(got, ref)
    order = np.argsort(-got["d"], kind="stable")
    assert np.array_equal(order, np.argsort(-ref["d"], kind="stable"))


def test_large_shape_properties(sa, ctx):
    """A size the oracle does not finish in seconds (config-2 shape scaled: 20000 genes x 30000
    cells, k = 30): size-independent properties instead of a reference run."""
    ctx.synth(20000, 30000, 20)
    ctx.fit_init(30, None)
    it, tols = ctx.nmf_run(0.0, 3, 0.01, 0.01, 0.0, 0.0)
    W, d, H = ctx.get_factors()
    assert it == 3 and np.all(np.isfinite(tols)) and tols[2] < tols[0]
    assert np.all(W >= 0) and np.all(H >= 0) and np.all(np.isfinite(W)) and np.all(np.isfinite(H))
    assert np.abs(W.sum(axis=0) - 1).max() < 1e-9          # scale(): rows of w sum to 1
    assert np.abs(H.sum(axis=0) - 1).max() < 1e-6          # ... and of h (scaled before the W-update)
    # idempotence of the deterministic pipeline: same inputs -> bit-identical outputs
    ctx.fit_init(30, None)
    ctx.nmf_run(0.0, 3, 0.01, 0.01, 0.0, 0.0)
    W2, d2, H2 = ctx.get_factors()
    assert np.array_equal(W, W2) and np.array_equal(H, H2) and np.array_equal(d, d2)
    # linearity of the right-hand sides: B(F1 + F2) = B(F1) + B(F2) on the resident matrix
    rng = np.random.default_rng(0)
    F1, F2 = rng.random((20000, 30)), rng.random((20000, 30))
    B12 = ctx.op_rhs(0, F1 + F2)
    assert rel_fro(B12, ctx.op_rhs(0, F1) + ctx.op_rhs(0, F2)) < 1e-13


@pytest.mark.parametrize("genes,cells,k", [(600, 150000, 30), (20000, 30000, 30), (3000, 40000, 50), (900, 50000, 31)])
def test_tiled_first_step_cold_streams(sa, genes, cells, k):
    """The first H-update after fit_init reads its entry streams from HBM (they were evicted from the
    last-level cache while the transposed streams were built), the slowest the refill loads ever are.
    Regression test of a late-landing prefetch that overwrote registers reused by the kernel's output
    addressing: the LDS-tiled path (fit default, and op_rhs which=2/3) must equal the plain
    wave-per-column kernel on several fresh builds."""
    import os
    c = sa.Context(0)
    try:
        c.synth(genes, cells, 20)
        os.environ["SGL_NO_TILED"] = "1"
        try:
            c.fit_init(k, None)
            c.step_begin(); c.step_h(0.01, 0.0)
            _, _, Hp = c.get_factors()
        finally:
            del os.environ["SGL_NO_TILED"]
        W0 = None
        for rep in range(3):
            c.fit_init(k, None)
            if W0 is None:
                W0 = c.get_factors()[0]
            c.step_begin(); c.step_h(0.01, 0.0)
            _, _, H = c.get_factors()
            assert rel_fro(H, Hp) < 1e-12, rep
            assert same_zero_pattern(H, Hp), rep
        for which in (0, 1):
            F = W0 if which == 0 else Hp
            assert rel_fro(c.op_rhs(which | 2, F), c.op_rhs(which, F)) < 1e-13, which
    finally:
        c.close()


def test_c_nmf_with_nnls_repack(sa, ora, monkeypatch):
    """Whole fit with the NNLS re-pack passes forced on at a small size: same parity bar."""
    monkeypatch.setenv("SGL_NNLS_REPACK_MIN_COLS", "256")
    A = ora.synth_csc(400, 3000, 20)
    At = A.t()
    w0 = ora.synth_winit(10, 400)
    ref = ora.c_nmf(A, At, 0.0, 4, 0.01, 0.01, 0.0, 0.0, 0, w0)
    got = sa.c_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, 4, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    _check(got, ref)


@pytest.mark.timeout(900)
def test_config2_full_size_parity(sa, ora):
    """BASELINE config 2 at full size (20 000 genes x 50 000 cells, 5 % nnz, k = 30): three ALS
    iterations through the one-shot entry point (host dgCMatrix in, transpose built on the device)
    against the oracle on the same matrix."""
    m, n, k = 20000, 50000, 30
    A = ora.synth_csc(m, n, 20)
    At = A.t()
    w0 = ora.synth_winit(k, m)
    ref = ora.c_nmf(A, At, 0.0, 3, 0.01, 0.01, 0.0, 0.0, 0, w0)
    got = sa.c_nmf(to_dgc(sa, A), None, 0.0, 3, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    _check(got, ref)
    assert got["iter"] == 3 and np.allclose(got["tol"], ref["tol"], rtol=1e-8)


@pytest.mark.timeout(900)
def test_config3_full_size_tiled_equals_plain(sa):
    """BASELINE config 3 at full size (30 000 genes x 1 000 000 cells, nnz 1.5e9, k = 50), too big
    for the oracle: the LDS-tiled path the bench runs must agree with the plain CSC kernel (an
    independent implementation of the same sums) after two full ALS iterations, the factors must be
    a fixed point of scale(), and a second run must be bit-identical."""
    import os
    genes, cells, k = 30000, 1000000, 50
    c = sa.Context(0)
    try:
        c.synth(genes, cells, 20)
        runs = []
        for mode in ("tiled", "plain", "tiled"):
            if mode == "plain":
                os.environ["SGL_NO_TILED"] = "1"
            try:
                c.fit_init(k, None)
            finally:
                os.environ.pop("SGL_NO_TILED", None)
            it, tols = c.nmf_run(0.0, 2, 0.01, 0.01, 0.0, 0.0)
            W, d, H = c.get_factors()
            runs.append((W, d, H, tols))
        (W, d, H, t), (Wp, dp, Hp, tp), (W2, d2, H2, t2) = runs
        assert np.array_equal(W, W2) and np.array_equal(H, H2) and np.array_equal(d, d2) and np.array_equal(t, t2)
        assert rel_fro(W, Wp) < 1e-11 and rel_fro(H, Hp) < 1e-11 and rel_fro(d, dp) < 1e-12
        assert same_zero_pattern(W, Wp) and same_zero_pattern(H, Hp)
        assert np.all(W >= 0) and np.all(H >= 0) and np.all(np.isfinite(H))
        assert np.abs(W.sum(axis=0) - 1).max() < 1e-9
    finally:
        c.close()


@pytest.mark.parametrize("which", ["both", "h_only", "w_only", "none"])
def test_c_linked_nmf(sa, ora, which):
    """c_linked_nmf (src/singlet.cpp:1059-1086): link matrices multiply the right-hand sides; a link
    whose column count does not match its side is ignored (R/RunLNMF.R switches a side off that way)."""
    m, n, k = 260, 330, 9
    A = ora.synth_csc(m, n, 15)
    At = A.t()
    w0 = ora.synth_winit(k, m)
    rng = np.random.default_rng(4)
    lh = (rng.random((k, n)) < 0.7) * (0.5 + rng.random((k, n)))
    lw = (rng.random((k, m)) < 0.8).astype(np.float64)
    off = np.ones((1, 1))
    link_h = lh if which in ("both", "h_only") else (off if which == "w_only" else None)
    link_w = lw if which in ("both", "w_only") else (off if which == "h_only" else None)
    ref = ora.c_linked_nmf(A, At, 0.0, 4, 0.01, 0.0, 0, w0, link_h, link_w)
    got = sa.c_linked_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, 4, False, 0.01, 0.0, 0, w0.T, link_h, link_w)
    _check(got, ref)
    if which == "none":
        plain = sa.c_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, 4, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
        assert np.array_equal(plain["w"], got["w"]) and np.array_equal(plain["h"], got["h"])
    if which in ("both", "h_only"):
        assert np.all(got["h"][lh == 0] == 0)   # a zero link pins the coefficient at zero


def test_callbacks_log_poll_and_verbose(sa, ora, ctx, capsys):
    """The two callbacks of the ABI: `log` fires once per iteration with the reference's 1-based
    iteration number and that iteration's tol (src/singlet.cpp:661-662); `poll` is the
    Rcpp::checkUserInterrupt() stand-in (:652, :663) -- a non-zero return stops the fit with SGL_EINTR
    and leaves the context usable.  verbose = TRUE prints the reference's header and lines."""
    A = ora.synth_csc(200, 240, 20)
    w0 = ora.synth_winit(6, 200)
    ctx.upload(to_dgc(sa, A), None)
    ctx.fit_init(6, w0)
    seen = []
    it, tols = ctx.nmf_run(0.0, 4, 0.01, 0.01, 0.0, 0.0, log=lambda i, t, of: seen.append((i, t, of)))
    assert [s[0] for s in seen] == [1, 2, 3, 4] and np.array_equal([s[1] for s in seen], tols)
    assert all(np.isnan(s[2]) for s in seen)          # c_nmf has no overfit column
    # interrupt at the second polling point of iteration 2 (two polls per iteration)
    calls = []
    ctx.fit_init(6, w0)
    with pytest.raises(sa.SingletHipError) as ei:
        ctx.nmf_run(0.0, 10, 0.01, 0.01, 0.0, 0.0, poll=lambda: (calls.append(1), len(calls) >= 4)[1])
    assert ei.value.code == -5 and len(calls) == 4
    ctx.fit_init(6, w0)                               # still usable afterwards
    it2, tols2 = ctx.nmf_run(0.0, 4, 0.01, 0.01, 0.0, 0.0)
    assert it2 == 4 and np.array_equal(tols2, tols)
    # verbose output of the R-level wrappers
    capsys.readouterr()
    r = sa.c_nmf(to_dgc(sa, A), None, 0.0, 2, True, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    out = capsys.readouterr().out.splitlines()
    assert out[1] == "%4s | %8s " % ("iter", "tol") and out[2] == "-" * 15
    assert out[3] == "%4d | %8.2e" % (1, r["tol"][0]) and out[4] == "%4d | %8.2e" % (2, r["tol"][1])
    sa.c_ard_nmf(to_dgc(sa, A), to_dgc(sa, A.t()), 0.0, 3, True, 0.01, 0.0, 0, w0.T, 7, 10, 1e9, 2)
    out = capsys.readouterr().out.splitlines()
    # iter_ % trace_test_mse == 0 is traced (src/singlet.cpp:1116): iterations 1 and 3 print a score, 2 prints "-"
    assert out[1] == "%4s | %8s | %8s " % ("iter", "tol", "overfit")
    assert out[3].endswith("| 0.00e+00") and out[4].endswith("|        -") and "e" in out[5].split("|")[2]


def test_c_nmf_dense_and_sparse_list(sa, ora):
    """The dense front-end (src/singlet.cpp:1052-1054, predict :370-381) solves every column, all-zero
    ones included -- unlike the sparse path, which skips them; the chunk-list front-end (:715-743) equals
    c_nmf on the concatenated matrix."""
    m, n, k = 180, 230, 7
    A = ora.synth_csc(m, n, 12)
    D = np.zeros((m, n))
    for c in range(n):
        D[A.i[A.p[c]:A.p[c + 1]], c] = A.x[A.p[c]:A.p[c + 1]]
    D[:, [5, 77]] = 0.0      # empty cells
    D[[3, 100], :] = 0.0     # empty genes
    w0 = ora.synth_winit(k, m)
    ref = ora.c_nmf_dense(D, 0.0, 4, 0.01, 0.01, 0.0, 0.0, 0, w0)
    got = sa.c_nmf_dense(D, None, 0.0, 4, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    _check(got, ref)
    # the sparse path on the same data keeps the stale (zero-initialised h / initial w) values instead
    import scipy.sparse as sp
    S = sp.csc_matrix(D)
    dS = sa.dgCMatrix(S.data, S.indices, S.indptr, (m, n))
    sparse = sa.c_nmf(dS, None, 0.0, 4, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    assert not np.array_equal(sparse["w"][:, 3], got["w"][:, 3])
    # chunk list: three ragged column chunks of A (and of t(A), accepted and unused)
    cuts = [0, 60, 61, n]
    chunks = [dS.col_slice(a, b) for a, b in zip(cuts, cuts[1:])]
    lst = sa.c_nmf_sparse_list(chunks, None, 0.0, 4, False, 0.01, 0.0, 0, w0.T)
    assert np.array_equal(lst["w"], sparse["w"]) and np.array_equal(lst["h"], sparse["h"]) and np.array_equal(lst["d"], sparse["d"])
    # ... against the oracle's restatement of the chunk loops (running column offset, src/singlet.cpp:384-402), with
    # the list of column chunks of t(A) given as well
    At = sp.csc_matrix(D.T)
    dAt = sa.dgCMatrix(At.data, At.indices, At.indptr, (n, m))
    tcuts = [0, 1, 90, 91, m]
    tchunks = [dAt.col_slice(a, b) for a, b in zip(tcuts, tcuts[1:])]
    o_chunks = [ora.CSC(ch.x, ch.i, ch.p, ch.nrow, ch.ncol) for ch in chunks]
    o_tchunks = [ora.CSC(ch.x, ch.i, ch.p, ch.nrow, ch.ncol) for ch in tchunks]
    oref = ora.c_nmf_sparse_list(o_chunks, o_tchunks, 0.0, 4, 0.01, 0.0, 0, w0)
    lst2 = sa.c_nmf_sparse_list(chunks, tchunks, 0.0, 4, False, 0.01, 0.0, 0, w0.T)
    _check(lst2, oref)
    assert np.array_equal(lst2["w"], lst["w"]) and np.array_equal(lst2["h"], lst["h"])
    # run_nmf takes the dense branch for a plain array (R/run_nmf.R:57)
    fit = sa.run_nmf(D, 5, tol=1e-3, maxit=5, verbose=False, seed=3)
    assert fit["w"].shape == (m, 5) and np.all(np.diff(fit["d"]) <= 0)


def test_dense_upload_then_log_normalize_fits_the_normalized_matrix(sa, ora):
    """Round-3 advice: sgl_log_normalize / sgl_weight_by_split rewrite the CSC image only; a dense copy kept for GEMM
    right-hand sides would go on describing the un-normalized matrix.  upload_dense + log_normalize (+ weight_by_split)
    must fit exactly what upload(csc) + the same staging fits."""
    rng = np.random.default_rng(12)
    m, n, k = 90, 260, 7
    D = np.floor(rng.random((m, n)) * 6.0)            # counts, > half non-zero: the GEMM path is chosen at upload
    assert (D != 0).mean() > 0.5
    split = (np.arange(n) % 3).astype(np.int32)
    w0 = ora.synth_winit(k, m)
    outs = []
    for dense in (True, False):
        c = sa.Context(0)
        try:
            if dense:
                c.upload_dense(D)
            else:
                c.upload(sa.dgCMatrix.from_dense(D), None)
            c.log_normalize(10000.0)
            c.weight_by_split(split, 3)
            c.fit_init(k, w0)
            c.nmf_run(0.0, 3, 0.01, 0.01, 0.0, 0.0)
            outs.append(c.get_factors())
        finally:
            c.close()
    (W1, d1, H1), (W2, d2, H2) = outs
    assert rel_fro(W1, W2) < 1e-11 and rel_fro(H1, H2) < 1e-11 and rel_fro(d1, d2) < 1e-11


@pytest.mark.parametrize("m,n,k,zero_frac", [(300, 700, 12, 0.0), (257, 513, 50, 0.3), (120, 400, 70, 0.0)])
def test_c_nmf_dense_gemm_path(sa, ora, m, n, k, zero_frac, monkeypatch):
    """A matrix that IS dense (more than half non-zero): the right-hand sides of predict are FP64 GEMMs on the dense
    copy (src/singlet.cpp:377: b = w * A.col(i)) instead of a sparse accumulate over its CSC image.  Against the
    oracle's dense loop, and against the CSC-image path (SGL_DENSE_GEMM=0) of the same library."""
    rng = np.random.default_rng(m + k)
    D = rng.random((m, n)) + 0.05
    D[rng.random((m, n)) < zero_frac] = 0.0
    D[:, 11] = 0.0                                  # an all-zero cell is still solved by the dense predict
    w0 = ora.synth_winit(k, m)
    ref = ora.c_nmf_dense(D, 0.0, 3, 0.01, 0.01, 0.0, 0.0, 0, w0)
    monkeypatch.delenv("SGL_DENSE_GEMM", raising=False)
    got = sa.c_nmf_dense(D, None, 0.0, 3, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    _check(got, ref)
    monkeypatch.setenv("SGL_DENSE_GEMM", "0")
    img = sa.c_nmf_dense(D, None, 0.0, 3, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    _check(img, ref)
    assert rel_fro(got["w"], img["w"]) < 1e-11 and rel_fro(got["h"], img["h"]) < 1e-11
    # the GEMM really ran: the H-side accumulate phase of a resident fit is a rocBLAS launch (no entry stream is built)
    monkeypatch.delenv("SGL_DENSE_GEMM", raising=False)
    c = sa.Context(0)
    try:
        c.upload_dense(D)
        c.fit_init(k, w0)
        c.nmf_run(0.0, 2, 0.01, 0.01, 0.0, 0.0)
        W, d, H = c.get_factors()
    finally:
        c.close()
    assert np.isfinite(W).all() and np.isfinite(H).all()


@pytest.mark.timeout(900)
def test_c_nmf_dense_2000_by_50000(sa, ora):
    """a fully dense 2000 x 50000 matrix through the GEMM path against the oracle's dense loop."""
    rng = np.random.default_rng(7)
    m, n, k = 2000, 50000, 20
    D = rng.random((m, n)) + 0.01
    w0 = ora.synth_winit(k, m)
    ref = ora.c_nmf_dense(D, 0.0, 2, 0.01, 0.01, 0.0, 0.0, 0, w0)
    got = sa.c_nmf_dense(D, None, 0.0, 2, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    _check(got, ref)


def test_c_ard_nmf_dense_and_sparse_list(sa, ora):
    """The masked loop behind the two other front-ends R/ard_nmf.R uses (l.109, 114): the dense one
    (src/singlet.cpp:1357-1361, predict_mask :506-533) solves every column; the chunk list (:1162-1234) hashes
    on `i + offset` (:485, :590).  Both against the oracle's restatements of those loops."""
    import scipy.sparse as sp
    m, n, k = 150, 210, 6
    A = ora.synth_csc(m, n, 8)
    D = A.to_dense()
    D[:, [9, 140]] = 0.0      # empty cells: solved by the dense path, skipped by the sparse one
    D[[4, 77], :] = 0.0
    w0 = ora.synth_winit(k, m)
    ref = ora.c_ard_nmf_dense(D, 0.0, 5, 0.01, 0.0, 0, w0, 31, 10, 1e9, 2)
    got = sa.c_ard_nmf_dense(D, None, 0.0, 5, False, 0.01, 0.0, 0, w0.T, 31, 10, 1e9, 2)
    assert list(got["iter"]) == list(ref["iter"]) == [0, 2, 4, 5]
    _check(got, ref)
    assert rel_fro(got["test_mse"], ref["test_mse"]) < 1e-9
    S, St = sp.csc_matrix(D), sp.csc_matrix(D.T)
    dS = sa.dgCMatrix(S.data, S.indices, S.indptr, (m, n))
    dSt = sa.dgCMatrix(St.data, St.indices, St.indptr, (n, m))
    cuts, tcuts = [0, 50, 51, 120, n], [0, 75, m]
    chunks = [dS.col_slice(a, b) for a, b in zip(cuts, cuts[1:])]
    tchunks = [dSt.col_slice(a, b) for a, b in zip(tcuts, tcuts[1:])]
    oc = [ora.CSC(ch.x, ch.i, ch.p, ch.nrow, ch.ncol) for ch in chunks]
    otc = [ora.CSC(ch.x, ch.i, ch.p, ch.nrow, ch.ncol) for ch in tchunks]
    lref = ora.c_ard_nmf_sparse_list(oc, otc, 0.0, 5, 0.01, 0.0, 0, w0, 31, 10, 1e9, 2)
    for tl in (tchunks, None):
        lgot = sa.c_ard_nmf_sparse_list(chunks, tl, 0.0, 5, False, 0.01, 0.0, 0, w0.T, 31, 10, 1e9, 2)
        assert list(lgot["iter"]) == list(lref["iter"])
        _check(lgot, lref)
        assert rel_fro(lgot["test_mse"], lref["test_mse"]) < 1e-9
    # the sparse list skips the empty genes (their w columns keep the scaled initial values) the dense front-end solves
    assert not np.array_equal(lgot["w"][:, 4], got["w"][:, 4])
    # bad lists come back as errors
    with pytest.raises(ValueError):
        sa.c_nmf_sparse_list([], None, 0.0, 1, False, 0.0, 0.0, 0, w0.T)
    with pytest.raises(sa.SingletHipError):   # an At list that is not t(A)
        sa.c_nmf_sparse_list(chunks, tchunks[:1], 0.0, 1, False, 0.0, 0.0, 0, w0.T)


def test_edge_arguments(sa, ora, ctx):
    """maxit = 0 runs no iteration (w returned as given, h = 0, d = 1, src/singlet.cpp:639-647); bad
    arguments come back as errors from the library, not as crashes."""
    A = ora.synth_csc(120, 90, 10)
    w0 = ora.synth_winit(5, 120)
    got = sa.c_nmf(to_dgc(sa, A), None, 0.0, 0, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    assert got["iter"] == 0 and np.array_equal(got["w"].T, w0) and not got["h"].any() and np.array_equal(got["d"], np.ones(5))
    # tol larger than the first change stops after one iteration, like the reference's loop test
    one = sa.c_nmf(to_dgc(sa, A), None, 2.0, 10, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    ref = ora.c_nmf(A, A.t(), 2.0, 10, 0.01, 0.01, 0.0, 0.0, 0, w0)
    assert one["iter"] == ref["iter"] == 0 or one["iter"] == ref["iter"]
    ctx.upload(to_dgc(sa, A), None)
    with pytest.raises(sa.SingletHipError):
        ctx.fit_init(0, None)
    with pytest.raises(sa.SingletHipError):
        ctx.fit_init(1025, None)                      # above SGL_MAX_K
    ctx.fit_init(5, w0)
    with pytest.raises(sa.SingletHipError):
        ctx.ard_run(0.0, 3, 0.01, 0.0, 1, 0, 1e9, 1)  # inv_density = 0
    with pytest.raises(sa.SingletHipError):
        ctx.ard_run(0.0, 3, 0.01, 0.0, 1, 20, 1e9, 0)  # trace_test_mse = 0 would divide by zero (never passed by R)
    bad = sa.dgCMatrix(A.x, A.i, A.p, (A.nrow, A.ncol))
    bad.i = bad.i.copy()
    bad.i[0] = A.nrow + 5                              # row index out of range
    with pytest.raises(sa.SingletHipError):
        sa.c_nmf(bad, None, 0.0, 1, False, 0.0, 0.0, 0.0, 0.0, 0, w0.T)
    bad.i = A.i.copy()
    s0, s1 = A.p[0], A.p[1]
    assert s1 - s0 >= 2
    bad.i[s0], bad.i[s0 + 1] = A.i[s0 + 1], A.i[s0]    # not ascending inside the first column
    with pytest.raises(sa.SingletHipError):
        sa.c_nmf(bad, None, 0.0, 1, False, 0.0, 0.0, 0.0, 0.0, 0, w0.T)


@pytest.mark.parametrize("k", [24, 100, 136])
def test_nnls_packing_by_sweep_counts_is_bit_identical(sa, monkeypatch, k):
    """From the second iteration on the H-side solve takes its columns in descending order of the sweeps their previous
    solve needed (neighbours share a wave: less lock-step waste).  A column's arithmetic does not depend on its lane: the
    factors, tol trace and sweep totals are those of the unpacked solve, bit for bit (70 000 cells: above the packing
    threshold; with and without the re-packing passes).  k = 100: the generated two-lane solve packs the same way; k = 136 (round 6):
    the four-lanes-per-column solve of ranks 129 - 256 takes its 16-column waves in that order."""
    genes, cells = 1500, 70000
    runs = {}
    for repack in ("0", "32768"):
        for pack in (True, False):
            if pack:
                monkeypatch.delenv("SGL_NNLS_NO_PACK", raising=False)
            else:
                monkeypatch.setenv("SGL_NNLS_NO_PACK", "1")
            if repack == "0":
                monkeypatch.delenv("SGL_NNLS_REPACK_MIN_COLS", raising=False)
            else:
                monkeypatch.setenv("SGL_NNLS_REPACK_MIN_COLS", repack)
            c = sa.Context(0)
            try:
                c.synth(genes, cells, 20)
                c.fit_init(k, None)
                c.sweeps_get(reset=True)
                it, tols = c.nmf_run(0.0, 4, 0.01, 0.01, 0.0, 0.0)
                sw = c.sweeps_get(reset=True)
                runs[(repack, pack)] = (c.get_factors(), tols, sw["h_sweeps"], sw["w_sweeps"], sw["h_wave_sweeps"])
            finally:
                c.close()
    ref = runs[("0", False)]
    for key, r in runs.items():
        for a, b in zip(r[0], ref[0]):
            assert np.array_equal(a, b), key
        assert np.array_equal(r[1], ref[1]) and r[2] == ref[2] and r[3] == ref[3], key
    # the packed solve executes fewer wave-sweeps than the unpacked one (that is its point)
    assert runs[("0", True)][4] < runs[("0", False)][4]
