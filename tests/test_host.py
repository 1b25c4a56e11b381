"""Host-side mirror of the R drivers (rank search bookkeeping) on the CPU: the numerics are
replaced by synthetic error curves, so only the control flow of R/GetBestRank.R,
R/ard_nmf.R:95-160 and R/cross_validate_nmf.R:69-97 is exercised."""
import numpy as np
import pytest


def rows(k, rep, errs, iters=None):
    iters = iters or list(range(0, 5 * len(errs), 5))
    return [dict(k=k, rep=rep, test_error=e, iter=i, tol=1e-3) for e, i in zip(errs, iters)]


def test_GetBestRank_picks_min_of_last_iterations(sa):
    df = rows(2, 1, [0.5, 0.4]) + rows(4, 1, [0.45, 0.30]) + rows(8, 1, [0.40, 0.35])
    assert sa.GetBestRank(df) == 4


def test_GetBestRank_overfit_rank_is_excluded(sa):
    # at k = 8 the test error goes UP along the fit: (v2 - v1) / (v2 + v1) > tol -> max_rank = 8
    df = rows(2, 1, [0.5, 0.4]) + rows(4, 1, [0.45, 0.33]) + rows(8, 1, [0.30, 0.20, 0.29])
    assert sa.GetBestRank(df, 1e-4) == 4
    # with a huge tolerance nothing is excluded and k = 8 wins on its last iteration
    assert sa.GetBestRank(df, 10.0) == 8


def test_GetBestRank_running_minimum_and_replicates(sa):
    # v1 is a running minimum: 0.30 -> 0.20 -> (0.25 replaced by 0.20) vs v2 = 0.20, 0.25, 0.21
    df = rows(3, 1, [0.30, 0.20, 0.25, 0.21]) + rows(5, 1, [0.5, 0.1])
    assert sa.GetBestRank(df, 1e-4) == 2          # both k < max_rank fail: nothing left -> 2
    df = rows(3, 1, [0.4, 0.3]) + rows(6, 1, [0.3, 0.2]) + rows(3, 2, [0.4, 0.25]) + rows(6, 2, [0.5, 0.3])
    assert sa.GetBestRank(df) == 4                # floor(mean(6, 3))
    assert sa.GetBestRank(rows(7, 1, [0.3])) == 7  # single row table (the nrow(df) == 1 branch)


def _fake_c_ard(curve):
    """c_ard_nmf replacement: test error is a function of k only."""
    calls = []

    def fake(A, At, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density, thr, trace):
        k = np.asarray(w).shape[0]
        calls.append((k, seed, inv_density))
        e = curve(k)
        of = 0.0 if k <= curve.best else 2e-3      # beyond the optimum the fit "overfits"
        return dict(w=np.zeros((k, A.nrow)), d=np.ones(k), h=np.zeros((k, A.ncol)), test_mse=np.array([e * 1.1, e]),
                    iter=np.array([0, 4], dtype=np.int32), tol=np.array([1e-2, 1e-4]), score_overfit=np.array([0.0, of]))
    return fake, calls


def test_ard_nmf_rank_search_control_flow(sa, monkeypatch):
    from singlet_amd import api

    def curve(k):
        return 0.5 + 0.01 * abs(k - 11)
    curve.best = 11
    fake, calls = _fake_c_ard(curve)
    monkeypatch.setattr(api, "c_ard_nmf", fake)
    monkeypatch.setattr(api, "c_nmf", lambda A, At, tol, maxit, verbose, L1w, L1h, L2w, L2h, threads, w: dict(
        w=np.asarray(w), d=np.arange(np.asarray(w).shape[0], dtype=float), h=np.zeros((np.asarray(w).shape[0], A.ncol)),
        iter=1, tol=np.array([0.0])))
    A = sa.dgCMatrix.from_dense(np.eye(6))
    model = api.ard_nmf(A, k_init=2, k_max=40, n_replicates=2, verbose=0, learning_rate=1, seed=3, resident=False)
    ks = [c[0] for c in calls]
    # replicate 1 (traced by hand through R/ard_nmf.R:121-158): step doubling 2, 4, 8, 16 (overfit ->
    # k_max = 16), 12 (overfit -> k_max = 12), then bisection 6, 10, 9, 11 and stop (neighbours 10 / 12)
    assert ks[:9] == [2, 4, 8, 16, 12, 6, 10, 9, 11]
    # replicate 2 inherits the shrunken k_max = 12 (R mutates k_max across replicates): 2, 4, 8, then 16 > k_max
    assert ks[9:] == [2, 4, 8]
    assert all(c[2] == 20 for c in calls)                       # round(1 / 0.05)
    seeds = sorted({c[1] for c in calls})
    assert len(seeds) == 2 and seeds[1] - seeds[0] == 1         # test_seed + curr_rep
    cv = model["cv_data"]
    assert cv.columns() == ["k", "rep", "test_error", "iter", "tol", "overfit_score"]
    # final fit at the best rank, factors sorted by d descending (R/run_nmf.R:65-68)
    kbest = model["d"].shape[0]
    assert kbest == 9 == sa.GetBestRank(cv, 1e-3)                # floor(mean(11, 8))
    assert np.all(np.diff(model["d"]) <= 0) and model["w"].shape == (6, kbest)


def test_cross_validate_nmf_grid_and_columns(sa, monkeypatch):
    from singlet_amd import api

    def curve(k):
        return 1.0 / k
    curve.best = 100
    fake, calls = _fake_c_ard(curve)
    monkeypatch.setattr(api, "c_ard_nmf", fake)
    A = sa.dgCMatrix.from_dense(np.eye(5))
    df = api.cross_validate_nmf(A, [2, 3, 5], n_replicates=2, verbose=0, seed=1, resident=False)
    assert [c[0] for c in calls] == [2, 3, 5, 2, 3, 5]          # expand.grid(k, rep): k varies fastest
    assert df.columns() == ["k", "rep", "test_error", "iter", "tol"]   # no overfit_score (R/cross_validate_nmf.R:90)
    assert len(df) == 12 and sa.GetBestRank(df) == 5
    with pytest.raises(ValueError):
        api.cross_validate_nmf(A, [2], L1=1.0)


def test_cross_validate_nmf_replica_devices(sa, monkeypatch):
    """The replica sweep (one resident copy of A per device, fits pulled from a queue, largest rank first) returns
    the one-device table row for row, whatever device ran which fit; errors of a worker surface on the caller."""
    from singlet_amd import api
    import threading
    log = []

    class FakeFits:
        def __init__(self, A, device=0):
            self.device = device

        def __enter__(self):
            return self

        def __exit__(self, *a):
            pass

        def close(self):
            pass

        def c_ard_nmf(self, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density, overfit_threshold, trace_test_mse):
            k = w.shape[0]
            log.append((self.device, k, seed, threading.current_thread().name))
            if k == 13:
                raise RuntimeError("boom")
            return {"test_mse": np.array([1.0 / k + 1e-9 * (seed % 7), 0.5 / k]), "iter": np.array([0, 5]), "tol": np.array([0.1, 0.01])}

    monkeypatch.setattr(api, "_ResidentFits", FakeFits)
    A = sa.dgCMatrix.from_dense(np.eye(5))
    one = api.cross_validate_nmf(A, [2, 3, 5], n_replicates=2, verbose=0, seed=1)
    log.clear()
    many = api.cross_validate_nmf(A, [2, 3, 5], n_replicates=2, verbose=0, seed=1, devices=[0, 1, 2])
    assert list(one) == list(many) and many.columns() == ["k", "rep", "test_error", "iter", "tol"]
    assert sorted(k for _, k, _, _ in log) == [2, 2, 3, 3, 5, 5] and {d for d, _, _, _ in log} <= {0, 1, 2}
    assert all(name.startswith("singlet-replica-") for _, _, _, name in log)
    monkeypatch.setenv("SINGLET_REPLICA_GPUS", "2")
    assert api._replica_devices(None) == [0, 1] and api._replica_devices(3) == [0, 1, 2] and api._replica_devices([4]) == [4]
    with pytest.raises(RuntimeError, match="boom"):
        api.cross_validate_nmf(A, [2, 13], n_replicates=1, verbose=0, seed=1, devices=2)


def test_run_nmf_argument_plumbing(sa, monkeypatch):
    from singlet_amd import api
    seen = {}

    def fake(A, At, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w):
        seen.update(tol=tol, maxit=maxit, L1=(L1_w, L1_h), L2=(L2_w, L2_h), wshape=np.asarray(w).shape)
        k = np.asarray(w).shape[0]
        return dict(w=np.asarray(w), d=np.array([1.0, 3.0, 2.0])[:k], h=np.arange(k * A.ncol, dtype=float).reshape(k, -1),
                    iter=1, tol=np.array([0.0]))
    monkeypatch.setattr(api, "c_nmf", fake)
    A = sa.dgCMatrix.from_dense(np.eye(4))
    m = api.run_nmf(A, 3, L1=[0.02, 0.03], verbose=False, seed=0)
    assert seen["L1"] == (0.02, 0.03) and seen["L2"] == (0.0, 0.0) and seen["wshape"] == (3, 4)
    assert seen["tol"] == 1e-4 and seen["maxit"] == 100           # R/run_nmf.R:18 defaults
    assert list(m["d"]) == [3.0, 2.0, 1.0] and m["w"].shape == (4, 3) and m["factor_names"][0] == "NMF_1"
    assert np.array_equal(m["h"][0], np.arange(4, 8, dtype=float))   # row of the largest d first
    with pytest.raises(ValueError):
        api.project_model(A, np.ones((3, 3)))


def test_r_drivers_pick_the_list_and_dense_entry_points(sa, monkeypatch):
    """R/ard_nmf.R:45-90, 105-111, 172-178 and R/cross_validate_nmf.R:27-63, 72-78: `"list" %in% class(A)` -> the
    *_sparse_list wrappers, `class(A)[[1]] == "matrix"` -> the *_dense wrappers, else the dgCMatrix ones."""
    from singlet_amd import api

    def curve(k):
        return 0.5 + 0.01 * abs(k - 5)
    curve.best = 5
    seen = []

    def fake_ard(name):
        inner, calls = _fake_c_ard(curve)

        def f(A, At, *rest):
            seen.append((name, type(A).__name__))
            return inner(_Shape(A), At, *rest)
        return f

    class _Shape:   # what _fake_c_ard needs of its first argument
        def __init__(self, A):
            if isinstance(A, list):
                self.nrow, self.ncol = A[0].nrow, sum(a.ncol for a in A)
            elif isinstance(A, np.ndarray):
                self.nrow, self.ncol = A.shape
            else:
                self.nrow, self.ncol = A.nrow, A.ncol

    def fake_nmf(name, nargs):
        def f(A, At, *rest):
            assert len(rest) == nargs, (name, len(rest))
            seen.append((name, type(A).__name__))
            w = np.asarray(rest[-1])
            return dict(w=w, d=np.arange(w.shape[0], dtype=float), h=np.zeros((w.shape[0], _Shape(A).ncol)), iter=1, tol=np.array([0.0]))
        return f

    monkeypatch.setattr(api, "c_ard_nmf", fake_ard("c_ard_nmf"))
    monkeypatch.setattr(api, "c_ard_nmf_sparse_list", fake_ard("c_ard_nmf_sparse_list"))
    monkeypatch.setattr(api, "c_ard_nmf_dense", fake_ard("c_ard_nmf_dense"))
    monkeypatch.setattr(api, "c_nmf", fake_nmf("c_nmf", 9))
    monkeypatch.setattr(api, "c_nmf_sparse_list", fake_nmf("c_nmf_sparse_list", 7))   # (tol, maxit, verbose, L1, L2, threads, w)
    monkeypatch.setattr(api, "c_nmf_dense", fake_nmf("c_nmf_dense", 9))
    D = np.eye(6) + 1.0
    one = sa.dgCMatrix.from_dense(D)
    chunks = [sa.dgCMatrix.from_dense(D[:, :2]), sa.dgCMatrix.from_dense(D[:, 2:])]
    for A, tag in ((one, ""), (chunks, "_sparse_list"), (D, "_dense")):
        seen.clear()
        m = api.ard_nmf(A, k_init=2, k_max=12, verbose=0, seed=3, resident=False)
        names = {n for n, _ in seen}
        assert names == {"c_ard_nmf" + tag, "c_nmf" + tag}, names
        assert seen[-1][0] == "c_nmf" + tag and m["w"].shape[0] == 6
        seen.clear()
        df = api.cross_validate_nmf(A, [2, 3], n_replicates=2, verbose=0, seed=1, resident=False)
        assert [n for n, _ in seen] == ["c_ard_nmf" + tag] * 4 and len(df) == 8
    with pytest.raises(ValueError):   # "number of rows in all provided 'A' matrices are not identical"
        api.ard_nmf([one, sa.dgCMatrix.from_dense(np.eye(5))], verbose=0)
