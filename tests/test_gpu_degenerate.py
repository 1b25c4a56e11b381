"""The reference's degenerate masked solves (round 5).

predict_mask solves every column against a_i = AAt(w) - AAt(w[:, drawn rows]) (src/singlet.cpp:458-463); both products carry
the same 1e-15 ridge, so the diagonal is NOT regularised: a factor whose row of w is all zero (a dead factor), or whose
only non-zero entries lie in the column's drawn rows, gives a_ii = 0 exactly, b_i = 0, and nnls's `b(i) / a(i, i)` (l.233) is
0 / 0.  The NaN then spreads as the reference's own statements spread it: x_i and every later coordinate of the column
become NaN, the earlier ones keep the value of the first sweep, tol becomes NaN and the column stops (l.231); cor of a NaN
factor is NaN and ends the ALS loop (`tol_ > tol`, l.1107).  The HIP path must land where the reference lands: the same
entries non-finite, the finite ones to 1e-9, the same iteration vectors.  (A zero diagonal that rests on cancellation of
MANY terms depends on the summation order and is not pinned; the cases here are exact in any order.)"""
import numpy as np
import pytest

from conftest import to_dgc

pytestmark = pytest.mark.gpu
TOL = 1e-9


def _same_where_finite(got, ref, what):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, what
    fin = np.isfinite(ref)
    assert np.array_equal(np.isnan(got), np.isnan(ref)), "%s: NaN pattern differs (%d vs %d NaN)" % (what, np.isnan(got).sum(), np.isnan(ref).sum())
    assert np.array_equal(np.isinf(got), np.isinf(ref)), "%s: Inf pattern differs" % what
    if fin.any():
        den = np.linalg.norm(ref[fin])
        assert np.linalg.norm(got[fin] - ref[fin]) <= TOL * (den if den > 0 else 1.0), what
        assert np.array_equal(got[fin] == 0, ref[fin] == 0), "%s: zero pattern differs" % what


# ranks through every per-column solve: LDS triangles (NR = 1, 3), the global-Gram quad solve (NR = 4, 6, 7), the wave kernel
@pytest.mark.parametrize("k", [6, 44, 50, 85, 100, 140])
@pytest.mark.parametrize("mask_t", [False, True])
@pytest.mark.parametrize("case", ["dead", "single"])
def test_predict_mask_with_a_zero_diagonal(sa, ora, ctx, k, mask_t, case):
    """dead: a factor whose row of the factor matrix is all zero -- a_ii = 1e-15 - 1e-15 = 0 in EVERY column;
    single: a factor with ONE non-zero entry -- a_ii = 0 exactly in the columns that draw that row, 4.0 in the others, so
    degenerate and regular columns share a wave."""
    m, n = 150, 200
    A = ora.synth_csc(m, n, 10)
    rng = np.random.default_rng(5)
    f = k // 3
    M = A.t() if mask_t else A                  # H-update: columns = cells, factor matrix w (m x k); W-update: genes, h (n x k)
    F = rng.random((M.nrow, k))
    F[:, f] = 0.0
    if case == "single":
        F[17, f] = 2.0
    ref = ora.predict_mask(M, 99, 8, F, np.zeros((M.ncol, k)), 0.01, 0.0, 0, mask_t)
    bad = np.isnan(ref).any(axis=1)
    assert bad.any() and np.isfinite(ref).any() and not np.isinf(ref).any()      # the case does what it is built for
    if case == "single":
        assert (~bad).sum() > M.ncol // 2          # most columns do not draw row 17 and stay regular
    # in a degenerate column the coordinates below the factor keep their first-sweep values, it and all behind it are NaN
    some = np.where(bad)[0][0]
    assert np.isfinite(ref[some, :f]).all() and np.isnan(ref[some, f:]).all()
    ctx.upload(to_dgc(sa, A))
    ctx.fit_init(k, rng.random((m, k)) if mask_t else F)
    if not mask_t:
        ctx.step_h_masked(0.01, 0.0, 99, 8)
        _, _, got = ctx.get_factors()
    else:
        ctx.set_factors(w=np.zeros((m, k)), h=F)
        ctx.step_w_masked(0.01, 0.0, 99, 8)
        got, _, _ = ctx.get_factors()
    _same_where_finite(got, ref, "x")


@pytest.mark.parametrize("k", [85, 86])
def test_c_ard_nmf_ends_where_the_reference_ends_when_a_factor_dies(sa, ora, k):
    """Round 4 met this case (k = 85 on 220 x 260: after the first H-update one factor of h is all zero, the W-update
    divides 0 by 0 in every gene) and dropped it as 'not a test case': the reference returns NaN factors, two trace rows
    (iterations 0 and 1) and stops after ONE iteration -- so must this engine."""
    A = ora.synth_csc(220, 260, 20)
    At = A.t()
    w0 = ora.synth_winit(k, 220)
    ref = ora.c_ard_nmf(A, At, 0.0, 2, 0.01, 0.0, 0, w0, 77, 20, 1e-3, 2)
    assert np.isnan(ref["w"]).any() and np.isfinite(ref["h"]).all() and list(ref["iter"]) == [0, 1]
    got = sa.c_ard_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, 2, False, 0.01, 0.0, 0, w0.T, 77, 20, 1e-3, 2)
    assert list(got["iter"]) == list(ref["iter"])
    _same_where_finite(got["w"].T, ref["w"], "w")
    _same_where_finite(got["h"].T, ref["h"], "h")
    _same_where_finite(got["d"], ref["d"], "d")
    assert np.isnan(got["test_mse"]).all() and np.isnan(ref["test_mse"]).all()
    assert np.isnan(got["tol"]).all() and np.isnan(ref["tol"]).all()


def test_non_finite_input_is_refused(sa, ora):
    """The shared-Gram solves assume finite operands; a matrix or an initial w with NaN / Inf is refused at the door
    (the reference would return all-NaN factors)."""
    A = ora.synth_csc(60, 80, 5)
    x = A.x.copy()
    x[7] = np.nan
    w0 = ora.synth_winit(4, 60)
    with pytest.raises(sa.SingletHipError) as e:
        sa.c_nmf(sa.dgCMatrix(x, A.i, A.p, (60, 80)), None, 1e-4, 3, False, 0.0, 0.0, 0.0, 0.0, 0, w0.T)
    assert "non-finite" in str(e.value)
    wbad = w0.copy()
    wbad[3, 2] = np.inf
    with pytest.raises(sa.SingletHipError) as e:
        sa.c_nmf(to_dgc(sa, A), None, 1e-4, 3, False, 0.0, 0.0, 0.0, 0.0, 0, wbad.T)
    assert "non-finite" in str(e.value)
