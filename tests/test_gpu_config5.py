"""BASELINE config 5 (ard_nmf + cross_validate_nmf rank sweep on the 30 000-gene matrix) on ONE GPU:
the masked (c_ard_nmf) path at the real gene count and >= 200 000 cells, where the CPU oracle cannot
follow, checked through properties; oracle parity on a slice of the same matrix that it can follow;
and the resident sweep (A uploaded once for the whole (rank, replicate) grid) against the one-shot
calls the R drivers make (R/ard_nmf.R:95-160, R/cross_validate_nmf.R:69-97)."""
import numpy as np
import pytest

from conftest import rel_fro, to_dgc

pytestmark = pytest.mark.gpu

GENES, CELLS = 30000, 200000
SEED, INV = 4711, 20


@pytest.fixture(scope="module")
def big(sa):
    c = sa.Context(0)
    c.synth(GENES, CELLS, 20)
    yield c
    c.close()


def test_mask_draw_rate_at_config5_offsets(big):
    """draw(cell, gene) at inv_density 20 marks ~5 % of the entries, also for cell indices near 1e6."""
    for cell0 in (0, 199000, 999000):
        m = big.op_mask(SEED, INV, cell0, 1000, GENES)
        rate = m.mean()
        assert abs(rate - 0.05) < 0.001, (cell0, rate)
        per_cell = m.mean(axis=1)
        assert per_cell.min() > 0.04 and per_cell.max() < 0.06


@pytest.mark.parametrize("k,iters", [(10, 3), (50, 3), (100, 2)])
def test_ard_run_at_config5_shape_is_finite_and_bit_reproducible(big, k, iters):
    res = []
    for _ in range(2):
        big.fit_init(k, None)
        r = big.ard_run(0.0, iters, 0.01, 0.0, SEED, INV, 1e9, 1)
        W, d, H = big.get_factors(h=False)
        res.append((r, W, d))
    (r0, W0, d0), (r1, W1, d1) = res
    assert r0["n_iter"] == iters and list(r0["iter"]) == list(range(iters))
    assert np.all(np.isfinite(r0["test_mse"])) and np.all(r0["test_mse"] > 0) and np.all(r0["test_mse"] < 10)
    assert np.all(np.diff(r0["test_mse"]) < 0)            # the first iterations of a fit reduce the test error
    assert np.all(np.isfinite(W0)) and np.all(W0 >= 0) and np.all(d0 > 0)
    assert abs(W0.sum(axis=0) - 1.0).max() < 1e-9         # scale(w): every factor sums to 1 over the genes
    # same inputs, same kernels, fixed-order reductions: bit-identical
    assert np.array_equal(r0["test_mse"], r1["test_mse"]) and np.array_equal(W0, W1) and np.array_equal(d0, d1)
    assert np.array_equal(r0["tol"], r1["tol"])


def test_entry_streams_survive_fit_reinit(big):
    """fit_init keeps the entry streams (and their buffers) of the resident matrix: same rank -> no rebuild, other
    rank -> rebuilt in the same allocations, other mask seed -> only the masked value array is refilled.  Every
    combination must give the bits of a fresh context."""
    def fit(ctx, k, seed):
        ctx.fit_init(k, None)
        r = ctx.ard_run(0.0, 2, 0.01, 0.0, seed, INV, 1e9, 1)
        W, d, _ = ctx.get_factors(h=False)
        return r["test_mse"].copy(), W.copy(), d.copy()

    seq = [(12, SEED), (12, SEED + 1), (30, SEED), (12, SEED)]
    got = [fit(big, k, s) for k, s in seq]
    assert np.array_equal(got[0][0], got[3][0]) and np.array_equal(got[0][1], got[3][1])   # back to the first (k, seed)
    assert not np.array_equal(got[0][0], got[1][0])                                          # another mask is another fit
    import singlet_amd as sa_mod
    fresh = sa_mod.Context(0)
    try:
        fresh.synth(GENES, CELLS, 20)
        # a context that never saw another rank or seed
        f1 = fit(fresh, 12, SEED + 1)
    finally:
        fresh.close()
    assert np.array_equal(f1[0], got[1][0]) and np.array_equal(f1[1], got[1][1]) and np.array_equal(f1[2], got[1][2])


def test_mask_lists_of_recent_seeds_are_kept(sa, monkeypatch):
    """A rank search refits one matrix with the seeds seed + 1 .. seed + n_replicates at rank after rank
    (R/ard_nmf.R:95-160), and the mask is a function of (seed, inv_density, cell, gene) alone: the lists of the last three
    masks stay resident and a fit under one of them hashes nothing.  The fits are the bits of a context that keeps none."""
    def sweep(ctx):
        out, builds = [], []
        for k, seed in [(8, SEED + 1), (8, SEED + 2), (8, SEED + 3), (14, SEED + 1), (14, SEED + 2), (14, SEED + 3),
                        (14, SEED + 4), (14, SEED + 2), (14, SEED + 1)]:
            ctx.fit_init(k, None)
            r = ctx.ard_run(0.0, 2, 0.01, 0.0, seed, INV, 1e9, 1)
            W, d, _ = ctx.get_factors(h=False)
            out.append((r["test_mse"].copy(), W.copy(), d.copy()))
            builds.append(ctx.layout_builds()[2:])
        return out, builds

    def run():
        c = sa.Context(0)
        try:
            c.synth(GENES, 20000, 20)
            return sweep(c)
        finally:
            c.close()

    kept, b_kept = run()
    # seeds 1, 2, 3 are hashed once each, the second rank reuses them; seed 4 pushes out the oldest (seed 1 -- 2 was used
    # after it); 2 is still there, 1 is hashed again
    assert b_kept == [(1, 1), (2, 2), (3, 3), (3, 3), (3, 3), (3, 3), (4, 4), (4, 4), (5, 5)], b_kept
    monkeypatch.setenv("SGL_MASK_KEEP", "0")
    import subprocess, sys, pickle, os, textwrap
    # (the switch is read once per process: the keep-nothing run is a child)
    code = textwrap.dedent("""
        import pickle, sys, numpy as np
        sys.path.insert(0, %r)
        import singlet_amd as sa
        c = sa.Context(0); c.synth(%d, 20000, 20)
        out = []
        for k, seed in [(8, %d), (14, %d), (14, %d)]:
            c.fit_init(k, None); r = c.ard_run(0.0, 2, 0.01, 0.0, seed, %d, 1e9, 1); W, d, _ = c.get_factors(h=False)
            out.append((r["test_mse"].copy(), W.copy(), d.copy(), c.layout_builds()[2:]))
        sys.stdout.buffer.write(pickle.dumps(out))
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), GENES, SEED + 1, SEED + 1, SEED + 2, INV)
    raw = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, timeout=600).stdout
    none = pickle.loads(raw[raw.index(b"\x80"):])
    assert [n[3] for n in none] == [(1, 1), (1, 1), (2, 2)]          # only the running mask is there
    for got, ref in ((kept[0], none[0]), (kept[3], none[1]), (kept[7], none[2])):
        assert all(np.array_equal(a, b) for a, b in zip(got, ref[:3]))


def test_config5_at_its_full_size_on_one_gpu(sa):
    """BASELINE config 5's matrix (30 000 genes x 1 000 000 cells, 1.5e9 non-zeros) fits one MI355X with both
    orientations, both entry streams and their masked value arrays (~110 GB): masked fits at the ends of the rank
    range run, reduce the test error, and repeat bit for bit; a re-init at the same rank reuses the streams."""
    c = sa.Context(0)
    try:
        c.synth(GENES, 1000000, 20)
        for k, iters in ((10, 3), (50, 2)):
            res = []
            for rep in range(2):
                c.fit_init(k, None)
                builds = c.layout_builds()
                r = c.ard_run(0.0, iters, 0.01, 0.0, SEED, INV, 1e9, 1)
                W, d, _ = c.get_factors(h=False)
                res.append((r["test_mse"].copy(), W.copy(), d.copy(), builds))
            (m0, W0, d0, b_first), (m1, W1, d1, b_again) = res
            assert np.all(np.isfinite(m0)) and np.all(np.diff(m0) < 0) and 0.1 < m0[-1] < 0.5
            assert abs(W0.sum(axis=0) - 1.0).max() < 1e-9 and np.all(d0 > 0)
            assert np.array_equal(m0, m1) and np.array_equal(W0, W1) and np.array_equal(d0, d1)
            # counted, not timed: hipMalloc / hipFree of the per-fit buffers cost 0.1 - 0.5 s at this size, box to box
            assert b_again[:2] == b_first[:2], "re-init at an unchanged rank rebuilt the entry streams %r -> %r" % (b_first, b_again)
    finally:
        c.close()


def test_never_drawn_mask_reduces_to_the_plain_fit(big):
    """With a divisor no hash value is a multiple of, predict_mask is predict up to the 1e-15 ridge the
    downdate cancels (src/singlet.cpp:461-462): the masked path (plain CSC kernel + per-column Grams +
    wave NNLS) must then reproduce the LDS-tiled / lane-NNLS path of c_nmf at the full size."""
    k = 20
    big.fit_init(k, None)
    big.nmf_run(0.0, 2, 0.01, 0.01, 0.0, 0.0)
    Wp, dp, Hp = big.get_factors()
    big.fit_init(k, None)
    r = big.ard_run(0.0, 2, 0.01, 0.0, SEED, (1 << 63) - 25, 1e9, 1)   # odd 63-bit divisor: never divides a hash here
    Wm, dm, Hm = big.get_factors()
    assert list(r["test_mse"]) == [0.0, 0.0]             # empty test set: losses are 0 by definition (l.563)
    assert rel_fro(Wm, Wp) < 1e-9 and rel_fro(Hm, Hp) < 1e-9 and rel_fro(dm, dp) < 1e-9


@pytest.mark.parametrize("k", [10, 50])
def test_oracle_parity_on_a_slice_with_all_genes(sa, ora, k):
    """The first 600 cells of the same synthetic matrix, all 30 000 genes, against the CPU restatement."""
    n = 600
    A = ora.synth_csc(GENES, n, 20)
    At = A.t()
    w0 = ora.synth_winit(k, GENES)
    ref = ora.c_ard_nmf(A, At, 0.0, 3, 0.01, 0.0, 0, w0, SEED, INV, 1e9, 1)
    got = sa.c_ard_nmf(to_dgc(sa, A), to_dgc(sa, At), 0.0, 3, False, 0.01, 0.0, 0, w0.T, SEED, INV, 1e9, 1)
    assert list(got["iter"]) == list(ref["iter"])
    assert rel_fro(got["test_mse"], ref["test_mse"]) < 1e-9
    assert rel_fro(got["w"].T, ref["w"]) < 1e-9 and rel_fro(got["h"].T, ref["h"]) < 1e-9


def test_resident_sweep_equals_one_shot_calls(sa, ora):
    """cross_validate_nmf / ard_nmf with A uploaded once return exactly what the per-call uploads return."""
    A = to_dgc(sa, ora.synth_csc(400, 900, 10))
    kw = dict(n_replicates=2, maxit=6, verbose=0, trace_test_mse=2, seed=5)
    a = sa.cross_validate_nmf(A, [3, 5, 8], resident=True, **kw)
    b = sa.cross_validate_nmf(A, [3, 5, 8], resident=False, **kw)
    assert a.columns() == b.columns() == ["k", "rep", "test_error", "iter", "tol"]
    assert len(a) == len(b) and all(ra == rb for ra, rb in zip(a, b))      # bit-identical rows
    kw = dict(k_init=2, k_max=12, n_replicates=1, maxit=8, verbose=0, seed=7, tol_overfit=1e-3)
    ma = sa.ard_nmf(A, resident=True, **kw)
    mb = sa.ard_nmf(A, resident=False, **kw)
    assert all(ra == rb for ra, rb in zip(ma["cv_data"], mb["cv_data"])) and len(ma["cv_data"]) == len(mb["cv_data"])
    assert np.array_equal(ma["w"], mb["w"]) and np.array_equal(ma["h"], mb["h"]) and np.array_equal(ma["d"], mb["d"])


def test_replica_sweep_equals_the_one_device_sweep(sa, ora):
    """cross_validate_nmf(devices = ...): the grid dealt out over several resident copies of A (here three contexts
    on the one device of this box, one host thread each) returns the one-device table bit for bit."""
    A = to_dgc(sa, ora.synth_csc(400, 900, 10))
    kw = dict(n_replicates=3, maxit=6, verbose=0, trace_test_mse=2, seed=5)
    a = sa.cross_validate_nmf(A, [3, 5, 8, 12], **kw)
    b = sa.cross_validate_nmf(A, [3, 5, 8, 12], devices=[0, 0, 0], **kw)
    assert len(a) == len(b) == 4 * 3 * len([r for r in a if r["k"] == 3 and r["rep"] == 1]) and all(ra == rb for ra, rb in zip(a, b))


def test_rank_limit_is_checked_before_the_upload(sa, ora):
    """The reference has no rank limit; the library's is 1024 (generic kernels above 128), and it is checked before
    anything is uploaded, with the limit in the message."""
    A = ora.synth_csc(200, 150, 10)
    w0 = np.ones((200, 1025))
    with pytest.raises(sa.SingletHipError) as e:
        sa.c_ard_nmf(to_dgc(sa, A), None, 0.0, 2, False, 0.01, 0.0, 0, w0.T, 1, 20, 1e9, 1)
    assert "1024" in str(e.value)
    with pytest.raises(sa.SingletHipError) as e:
        sa.c_nmf(to_dgc(sa, A), None, 0.0, 2, False, 0.01, 0.01, 0.0, 0.0, 0, w0.T)
    assert "1024" in str(e.value)


def test_one_shot_cache_keeps_the_matrix_resident(sa, ora, monkeypatch):
    """SINGLET_HIP_CACHE=1: the one-shot entry points reuse the resident matrix when the same host slots come
    again (what R's unchanged ard_nmf loop passes), notice a changed value at a sampled position, and give the
    results of the uncached calls bit for bit."""
    import time
    from singlet_amd import _lib
    A = ora.synth_csc(3000, 20000, 20)
    dA = to_dgc(sa, A)
    w5, w8 = ora.synth_winit(5, 3000), ora.synth_winit(8, 3000)
    plain5 = sa.c_ard_nmf(dA, None, 0.0, 3, False, 0.01, 0.0, 0, w5.T, 9, 20, 1e9, 1)
    plain8 = sa.c_nmf(dA, None, 0.0, 3, False, 0.01, 0.01, 0.0, 0.0, 0, w8.T)
    monkeypatch.setenv("SINGLET_HIP_CACHE", "1")
    try:
        t0 = time.perf_counter()
        c5 = sa.c_ard_nmf(dA, None, 0.0, 3, False, 0.01, 0.0, 0, w5.T, 9, 20, 1e9, 1)      # uploads, keeps
        t1 = time.perf_counter()
        c8 = sa.c_nmf(dA, None, 0.0, 3, False, 0.01, 0.01, 0.0, 0.0, 0, w8.T)               # reuses
        c5b = sa.c_ard_nmf(dA, None, 0.0, 3, False, 0.01, 0.0, 0, w5.T, 9, 20, 1e9, 1)     # reuses
        for key in ("w", "h", "d", "test_mse"):
            assert np.array_equal(c5[key], plain5[key]) and np.array_equal(c5b[key], plain5[key])
        for key in ("w", "h", "d"):
            assert np.array_equal(c8[key], plain8[key])
        # a value changed in place at a sampled position (the first entries are always sampled): re-uploaded
        dA.x[3] *= 2.0
        changed = sa.c_nmf(dA, None, 0.0, 3, False, 0.01, 0.01, 0.0, 0.0, 0, w8.T)
        monkeypatch.delenv("SINGLET_HIP_CACHE")
        ref = sa.c_nmf(dA, None, 0.0, 3, False, 0.01, 0.01, 0.0, 0.0, 0, w8.T)              # cache off: released, plain call
        assert np.array_equal(changed["w"], ref["w"]) and not np.array_equal(changed["w"], plain8["w"])
    finally:
        _lib.load().sgl_cache_release()


def test_r_driver_list_and_dense_branches_equal_the_one_matrix_result(sa, ora):
    """ard_nmf / cross_validate_nmf on a list of column chunks (R/ard_nmf.R:45-76 -> c_*_sparse_list) and on a dense
    matrix (R/ard_nmf.R:79-86 -> c_*_dense) against the same call on the one dgCMatrix: the list form joins the chunks into
    the same resident matrix (bit-identical); the dense form differs only by solving all-zero columns, of which this
    matrix has none (1e-9)."""
    O = ora.synth_csc(300, 700, 6)
    assert np.all(np.diff(O.p) > 0) and np.all(np.diff(O.t().p) > 0)      # no empty cell, no empty gene
    A = to_dgc(sa, O)
    cuts = [0, 150, 151, 480, 700]
    chunks = [sa.dgCMatrix(O.x[O.p[a]:O.p[b]], O.i[O.p[a]:O.p[b]], O.p[a:b + 1] - O.p[a], (300, b - a)) for a, b in zip(cuts[:-1], cuts[1:])]
    D = O.to_dense()
    kw = dict(n_replicates=2, maxit=6, verbose=0, trace_test_mse=2, seed=5)
    one = sa.cross_validate_nmf(A, [3, 6], **kw)
    lst = sa.cross_validate_nmf(chunks, [3, 6], **kw)
    den = sa.cross_validate_nmf(D, [3, 6], **kw)
    assert len(one) == len(lst) == len(den) and all(a == b for a, b in zip(one, lst))
    for a, b in zip(one, den):
        assert (a["k"], a["rep"], a["iter"]) == (b["k"], b["rep"], b["iter"])
        assert abs(a["test_error"] - b["test_error"]) <= 1e-9 * abs(a["test_error"])
    kw = dict(k_init=2, k_max=10, n_replicates=1, maxit=8, verbose=0, seed=7, tol_overfit=1e-3)
    m1, ml, md = sa.ard_nmf(A, **kw), sa.ard_nmf(chunks, **kw), sa.ard_nmf(D, **kw)
    assert [r["k"] for r in m1["cv_data"]] == [r["k"] for r in ml["cv_data"]] == [r["k"] for r in md["cv_data"]]
    assert np.array_equal(m1["w"], ml["w"]) and np.array_equal(m1["h"], ml["h"]) and np.array_equal(m1["d"], ml["d"])
    assert rel_fro(md["w"], m1["w"]) < 1e-9 and rel_fro(md["h"], m1["h"]) < 1e-9 and rel_fro(md["d"], m1["d"]) < 1e-9


def test_bench_ard_workload_prints_roofline_and_cpu_baseline():
    """`bench.py --workload ard` is how config 5 is measured (round 5): the grid's wall seconds, per rank the Gram downdate's
    FP64-MFMA TFLOP/s from its hipEvent phase and the drawn-pair count, and the oracle's c_ard_nmf timed on a cell slice --
    run here end to end on a small shape and held to the bench contract's keys."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "ard", "--genes", "3000", "--cells", "20000", "--ranks", "6,20,50",
                        "--replicates", "2", "--maxit", "3", "--trace", "2", "--cpu-sample-cells", "1500"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "per_rank"):
        assert key in d, key
    assert d["unit"] == "s" and d["higher_is_better"] is False and d["steps"] == 6 and d["value"] > 0
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 78.6 and 0 < rf["frac"] < 1 and rf["traffic"] is None
    assert [q["k"] for q in d["per_rank"]] == [6, 20, 50]
    for q in d["per_rank"]:
        dd = q["downdate"]
        # every (cell, gene) pair is drawn with probability 1 / 20, listed once per cell and once per gene
        assert abs(dd["pairs_per_iteration"] / (2 * 3000 * 20000 / 20.0) - 1.0) < 0.02
        assert dd["achieved_tflops"] > 0 and dd["ms_per_iteration"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and len(cb["per_rank"]) == 3 and "c_ard_nmf" in cb["sample"]
