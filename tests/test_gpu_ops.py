"""Parity of each HIP operator of the path against the CPU oracle, through the
C ABI (singlet_amd.Context wraps sgl_op_* one-to-one).  Integer work is
bit-exact; FP64 work is held to 1e-11 relative (the only arithmetic difference
is FMA contraction and reduction order), far inside the 1e-5 the north star asks."""
import numpy as np
import pytest

from conftest import rel_fro, to_dgc

pytestmark = pytest.mark.gpu

KATS = [((123, 0, 0), 0x692656729eb6707c), ((123, 1, 2), 0xd1c4f6746e22623f), ((123, 2, 1), 0xa171855b28d239a7),
        ((123, 999999, 29999), 0x3ffec3e4f2d21eea), ((2147483647, 5, 7), 0x3b72606ab1a2d601),
        ((1, 0, 1), 0x000112648b36e912)]


def test_rand_kats(ctx):
    for (s, i, j), v in KATS:
        assert int(ctx.op_rand(s, [i], [j])[0]) == v


def test_rand_bulk_bit_exact(ctx, ora):
    rng = np.random.default_rng(7)
    i = rng.integers(0, 2 ** 63, 5000, dtype=np.uint64)
    j = rng.integers(0, 2 ** 63, 5000, dtype=np.uint64)
    i[:4] = [0, 1, 2 ** 64 - 1, 2 ** 32]
    j[:4] = [2 ** 64 - 1, 0, 2 ** 64 - 1, 2 ** 31]
    for state in (0, 123, 2 ** 64 - 1):
        got = ctx.op_rand(state, i, j)
        exp = np.array([ora.rng_rand(state, int(a), int(b)) for a, b in zip(i, j)], dtype=np.uint64)
        assert np.array_equal(got, exp)


def _rng_ref():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rng_ref.npz"))
    return g, np.unpackbits(g["draw"], axis=-1)[..., :int(g["draw_shape"][-1])]


def test_rand_equals_the_reference_rng_class(ctx):
    """The device hash against 101 152 outputs of the reference's own `rng` class (src/singlet.cpp:6-114 compiled from
    the reference tree by oracle/make_ref.sh; fixture tests/golden/rng_ref.npz): bit-exact."""
    g, _ = _rng_ref()
    state, i, j, exp = g["state"], g["i"], g["j"], g["rand2"]
    for s in np.unique(state):
        sel = state == s
        assert np.array_equal(ctx.op_rand(int(s), i[sel], j[sel]), exp[sel])


def test_mask_equals_the_reference_rng_class(ctx):
    """The device mask (multiply-compare divisibility test) against draw(cell, gene, inv_density) grids of the
    reference's class: seven densities, cell offsets 0 and 999 000."""
    g, grids = _rng_ref()
    seed = int(g["draw_state"])
    nc, ng = grids.shape[2], grids.shape[3]
    for a, inv in enumerate(g["draw_inv_density"].tolist()):
        for b, c0 in enumerate(g["draw_cell0"].tolist()):
            assert np.array_equal(ctx.op_mask(seed, inv, c0, nc, ng), grids[a, b]), (inv, c0)


@pytest.mark.parametrize("inv_density", [20, 5, 1, 2, 3, 7, 64, 1000, 2 ** 20 + 7])
def test_mask_bit_exact(ctx, ora, inv_density):
    """draw = (rand % inv_density == 0): the device replaces the u64 modulo by a multiply-high test
    (sgl_divides); powers of two, small and large divisors, and d = 1 must all agree bit for bit."""
    ncells, ngenes = (37, 515) if inv_density < 1000 else (400, 3000)
    got = ctx.op_mask(42, inv_density, 1000, ncells, ngenes)
    exp = ora.rng_mask(42, 1000, ncells, ngenes, inv_density)
    assert np.array_equal(got, exp)


def _mask_gram_oracle(ora, F, G, ncols, seed, inv_density, mask_t, col_off, row_off):
    """G - (AAt(F[idx_c]) + 1e-15 I) per column (src/singlet.cpp:458-463), idx_c from the oracle's mask; G None: the raw sum."""
    nrow, k = F.shape
    if mask_t == 0:   # columns are cells, rows genes
        M = ora.rng_mask(seed, col_off, ncols, nrow + row_off, inv_density)[:, row_off:]
    else:             # columns are genes, rows cells
        M = ora.rng_mask(seed, row_off, nrow, ncols + col_off, inv_density)[:, col_off:].T
    out = np.empty((ncols, k, k))
    for c in range(ncols):
        Fs = F[M[c].astype(bool)]
        S = Fs.T @ Fs
        out[c] = S if G is None else G - (S + 1e-15 * np.eye(k))
    return out


@pytest.mark.parametrize("k", [1, 2, 5, 10, 16, 17, 18, 20, 21, 24, 25, 28, 30, 33, 36, 37, 40, 41, 44, 48, 50, 52, 53, 56, 57, 60, 64, 66, 69, 70, 72,
                               73, 76, 80, 84, 85, 88, 89, 92, 96, 98, 100, 104, 112, 128, 130])
@pytest.mark.parametrize("use_lists", [False, True, "quarter_mfma_remainder"])
def test_mask_gram_downdate(ctx, ora, k, use_lists, monkeypatch):
    """Per-column Gram downdates of predict_mask, by the hashing kernel and from the mask lists (every tile-set
    instance: full blocks, remainder rows on the VALU (round 6: 2 - 12 rows beside 1 - 4 tile rows) or as one to three quads of
    quarter-MFMAs (SGL_MASK_GRAM_NO_REMV=1: rounds 3 - 5), a partial last block, two-part tile sets; k = 130: the VALU kernel,
    which ignores the lists), both orientations with offsets, raw sums; several hundred drawn rows per column
    (the pipelined loop of the list kernel) down to none."""
    if use_lists == "quarter_mfma_remainder":
        monkeypatch.setenv("SGL_MASK_GRAM_NO_REMV", "1")
        use_lists = True
    else:
        monkeypatch.delenv("SGL_MASK_GRAM_NO_REMV", raising=False)
    rng = np.random.default_rng(500 + k)
    nrow, ncols = 1500, 7
    F = rng.random((nrow, k)) + 0.1
    G = ora.aat(rng.random((3 * k + 2, k)))
    for mask_t, inv, co, ro, raw in ((0, 5, 11, 0, False), (1, 4, 0, 23, False), (0, 3, 5, 0, True), (1, 1, 0, 0, False), (0, 700, 3, 0, False)):
        got = ctx.op_mask_gram(F, None if raw else G, ncols, 77, inv, mask_t, co, ro, use_lists)
        exp = _mask_gram_oracle(ora, F, None if raw else G, ncols, 77, inv, mask_t, co, ro)
        scale = np.abs(exp).max() + 1.0
        assert np.abs(got - exp).max() / scale < 1e-12, (mask_t, inv, raw)
        assert np.array_equal(got, got.transpose(0, 2, 1))


@pytest.mark.parametrize("nrow", [1, 3, 4, 5, 17, 64, 65, 250])
def test_mask_gram_lists_short_columns(ctx, ora, nrow):
    """Few rows: empty lists, one partial group, fewer groups than waves, the unpipelined tail only."""
    rng = np.random.default_rng(900 + nrow)
    for k in (7, 50, 100, 22, 43, 70, 90):
        F = rng.random((nrow, k)) + 0.1
        G = ora.aat(rng.random((2 * k, k)))
        for inv in (1, 2, 9):
            got = ctx.op_mask_gram(F, G, 9, 5, inv, 0, 2, 0, True)
            exp = _mask_gram_oracle(ora, F, G, 9, 5, inv, 0, 2, 0)
            assert np.abs(got - exp).max() / (np.abs(exp).max() + 1.0) < 1e-12, (k, inv)


@pytest.mark.parametrize("k,cols", [(1, 5), (8, 400), (16, 33), (30, 2000), (50, 777), (64, 300), (70, 129), (100, 50), (128, 300),
                                    (129, 200), (144, 3), (160, 1000), (161, 517), (192, 64), (193, 4099), (224, 333), (225, 130), (256, 2050),
                                    (257, 100), (300, 700), (520, 90)])
def test_gram(ctx, ora, k, cols):
    F = np.random.default_rng(k * 1000 + cols).random((cols, k))
    G = ctx.op_gram(F)
    E = ora.aat(F)
    assert rel_fro(G, E) < 1e-13
    assert np.array_equal(G, G.T)


def test_gram_transpose_detecting(ctx, ora):
    # asymmetric factor rows: a swapped MFMA C-layout would show up here
    F = np.zeros((40, 20))
    F[:, 3] = np.arange(40) + 1
    F[:, 17] = 1.0 / (np.arange(40) + 1)
    assert rel_fro(ctx.op_gram(F), ora.aat(F)) < 1e-14


@pytest.mark.parametrize("k", [1, 7, 30, 50, 64, 65, 100, 130, 257, 700])
def test_rhs_both_orientations(ctx, ora, sa, k):
    A = ora.synth_csc(300, 450, 20)
    At = A.t()
    ctx.upload(to_dgc(sa, A), to_dgc(sa, At))
    rng = np.random.default_rng(k)
    W = rng.random((A.nrow, k))
    H = rng.random((A.ncol, k))
    assert rel_fro(ctx.op_rhs(0, W), ora.rhs(A, W)) < 1e-14
    assert rel_fro(ctx.op_rhs(1, H), ora.rhs(At, H)) < 1e-14


@pytest.mark.parametrize("k", [1, 2, 7, 10, 16, 30, 31, 32, 33, 50, 64, 65, 70, 100, 127, 128, 129, 150, 200, 257, 600])
def test_rhs_tiled_kernel_all_ranks(ctx, ora, sa, k, monkeypatch):
    """The LDS-tiled accumulate (which = 2 / 3): four columns per LDS instruction up to k = 32 (256-byte tile rows),
    two up to k = 64, three / four quad passes over factor parts for 64 < k <= 128 (strided factor rows and outputs), odd ranks
    through the re-pitched staging.  1e-14 to the oracle as it runs by default (a matrix this small has its tile range cut
    over the CUs: partial sums added in range order); with the range whole (SGL_TILED_RANGES=1) bit-equal to the plain CSC
    kernel (same products, same order)."""
    A = ora.synth_csc(700, 900, 12)
    At = A.t()
    ctx.upload(to_dgc(sa, A), to_dgc(sa, At))
    rng = np.random.default_rng(k)
    W = rng.random((A.nrow, k))
    H = rng.random((A.ncol, k))
    for which, F, M in ((2, W, A), (3, H, At)):
        monkeypatch.delenv("SGL_TILED_RANGES", raising=False)
        got = ctx.op_rhs(which, F)
        assert rel_fro(got, ora.rhs(M, F)) < 1e-14
        monkeypatch.setenv("SGL_TILED_RANGES", "1")
        whole = ctx.op_rhs(which, F)
        assert rel_fro(whole, got) < 1e-14
        if k <= 64:
            assert np.array_equal(whole, ctx.op_rhs(which - 2, F))


@pytest.mark.parametrize("k", [2, 10, 16, 30, 32])
def test_rhs_tiled_quad_layout_equals_the_pair_layout(sa, ora, k, monkeypatch):
    """Ranks up to 32 on a matrix with several row tiles, column blocks and tile ranges (2500 x 3000, ~10 % non-zero):
    the quad layout (default) against the pair layout (SGL_TILED_NO_QUAD=1) and the plain kernel, bit for bit, and its
    stream is the one the layout query reports (128 columns per block, 632-row tiles)."""
    monkeypatch.setenv("SGL_TILED_RANGES", "1")   # bit for bit: the tile range whole (by default a matrix this small has it cut over the CUs)
    monkeypatch.setenv("SGL_TILED_FULL_TILES", "1")   # ... and LDS-sized tiles (it would get shorter ones)
    A = ora.synth_csc(2500, 3000, 10)
    At = A.t()
    rng = np.random.default_rng(100 + k)
    W, H = rng.random((A.nrow, k)), rng.random((A.ncol, k))
    out, lay = {}, {}
    for quad in (True, False):
        if quad:
            monkeypatch.delenv("SGL_TILED_NO_QUAD", raising=False)
        else:
            monkeypatch.setenv("SGL_TILED_NO_QUAD", "1")
        c = sa.Context(0)
        try:
            c.upload(to_dgc(sa, A), to_dgc(sa, At))
            out[quad] = (c.op_rhs(2, W), c.op_rhs(3, H), c.op_rhs(0, W), c.op_rhs(1, H))
            c.fit_init(k, ora.synth_winit(k, A.nrow))
            lay[quad] = c.layout_get()
        finally:
            c.close()
    for q in range(2):
        assert np.array_equal(out[True][q], out[False][q]) and np.array_equal(out[True][q], out[True][q + 2])
    assert rel_fro(out[True][0], ora.rhs(A, W)) < 1e-14 and rel_fro(out[True][1], ora.rhs(At, H)) < 1e-14
    assert lay[True]["A"]["col_blocks"] == (3000 + 127) // 128 and lay[False]["A"]["col_blocks"] == (3000 + 63) // 64
    assert lay[True]["A"]["tile_rows"] == 632 and lay[True]["A"]["tiles"] == 4


@pytest.mark.parametrize("k", [12, 50, 70])
def test_rhs_tiled_tail_split(sa, ora, k, monkeypatch):
    """More than 256 column groups: the workgroups run in rounds of 256 and only those of the last, partly filled round
    have their tile range cut into pieces (own compact slabs, summed in piece order).  140 000 cells x 2500 genes: 274 (137
    at k <= 32) groups of 512 (1024) columns; with and without the tail split (SGL_TILED_NO_TAIL) the sums agree to rounding,
    and both with the oracle."""
    quad = k <= 32 or 64 < k <= 128     # four columns per tuple: ranks up to 32, and the three / four passes of ranks 65 - 128 (round 5)
    A = ora.synth_csc(2500, 280000 if quad else 140000, 20)
    rng = np.random.default_rng(k)
    W = rng.random((A.nrow, k))
    out = {}
    monkeypatch.setenv("SGL_TILED_RANGES", "1")   # (below 1024 groups the uniform split would be chosen: keep the range whole)
    for tail in (True, False):
        if tail:
            monkeypatch.delenv("SGL_TILED_NO_TAIL", raising=False)
        else:
            monkeypatch.setenv("SGL_TILED_NO_TAIL", "1")
        c = sa.Context(0)
        try:
            c.upload(to_dgc(sa, A), None)
            out[tail] = c.op_rhs(2, W)
        finally:
            c.close()
    assert not np.array_equal(out[True], out[False])          # the split really ran (its columns are summed in another order)
    assert rel_fro(out[True], out[False]) < 1e-14
    sel = np.r_[0:300, A.ncol - 300:A.ncol]
    sub = ora.CSC(np.concatenate([A.x[A.p[c]:A.p[c + 1]] for c in sel]), np.concatenate([A.i[A.p[c]:A.p[c + 1]] for c in sel]),
                  np.concatenate([[0], np.cumsum([A.p[c + 1] - A.p[c] for c in sel])]), A.nrow, len(sel))
    assert rel_fro(out[True][sel], ora.rhs(sub, W)) < 1e-14


@pytest.mark.parametrize("k,ranges", [(10, 3), (30, 4), (50, 2), (50, 5)])
def test_rhs_tiled_split_tile_ranges(sa, ora, k, ranges, monkeypatch):
    """The tile range cut over blockIdx.y (what fills the chip when the columns are few: the W side of every config):
    ranges of floor / ceil of T / R tiles, partial slabs summed in range order.  The sums are those of the whole range up
    to rounding (1e-14 to the oracle), not bit for bit; forced here on a small matrix with SGL_TILED_RANGES."""
    monkeypatch.setenv("SGL_TILED_RANGES", str(ranges))
    A = ora.synth_csc(3300, 700, 10)      # 6 tiles of 632 rows (k <= 32), 9 of 408 at k = 50
    At = A.t()
    rng = np.random.default_rng(7 * k + ranges)
    W, H = rng.random((A.nrow, k)), rng.random((A.ncol, k))
    c = sa.Context(0)
    try:
        c.upload(to_dgc(sa, A), to_dgc(sa, At))
        got = c.op_rhs(2, W), c.op_rhs(3, H)
        c.fit_init(k, ora.synth_winit(k, A.nrow))
        lay = c.layout_get()
    finally:
        c.close()
    assert lay["A"]["tile_ranges"] == ranges and lay["At"]["tile_ranges"] == min(ranges, lay["At"]["tiles"])
    assert rel_fro(got[0], ora.rhs(A, W)) < 1e-14 and rel_fro(got[1], ora.rhs(At, H)) < 1e-14


@pytest.mark.parametrize("layout", ["quad", "pair"])
@pytest.mark.parametrize("shape", ["very_sparse", "dense_blocks", "one_pair_only"])
def test_rhs_tiled_pair_bookkeeping_extremes(ctx, ora, sa, shape, layout, monkeypatch):
    """The chunk loop of the tiled accumulate walks a byte queue of per-pair group counts (gen_acc_tiled.py): most
    counts zero (long skips, queue rotation over empty 64-bit words), counts near the byte's limit (whole columns
    dense inside a tile), and a chunk whose entries all belong to one pair.  Against the plain kernel and the oracle."""
    if layout == "pair":
        monkeypatch.setenv("SGL_TILED_NO_QUAD", "1")
    monkeypatch.setenv("SGL_TILED_RANGES", "1")   # bit for bit: the tile range whole
    monkeypatch.setenv("SGL_TILED_FULL_TILES", "1")   # counts near the byte's limit need LDS-sized tiles
    rng = np.random.default_rng({"very_sparse": 1, "dense_blocks": 2, "one_pair_only": 3}[shape])
    m, n, k = 2300, 200, 10                     # k = 10: tiles of 984 rows (the cap; 632 in the quad layout)
    D = np.zeros((m, n))
    if shape == "very_sparse":
        idx = rng.integers(0, m * n, size=150)  # a handful of entries: almost every (chunk, pair) is empty
        D.flat[idx] = rng.random(150) + 0.1
        D[:, 64:128] = 0                        # a whole wave block without entries
    elif shape == "dense_blocks":
        D[:, 3] = rng.random(m) + 0.1           # 984 entries per tile in one column: 246 groups for its pair
        D[:, 35] = rng.random(m) + 0.1          # ... and in its partner of the pair (3, 35)
        D[100:1100, 70] = rng.random(1000) + 0.1
        D[(rng.random((m, n)) < 0.02)] = 0.5
    else:
        D[:, 17] = (rng.random(m) < 0.5) * (rng.random(m) + 0.1)
        D[:, 49] = (rng.random(m) < 0.1) * (rng.random(m) + 0.1)   # pair (17, 49) of wave block 0 only
    A = ora.CSC(*_csc_from_dense(D))
    At = A.t()
    ctx.upload(to_dgc(sa, A), to_dgc(sa, At))
    W = rng.random((m, k))
    H = rng.random((n, k))
    for which, F, M in ((2, W, A), (3, H, At)):
        got, plain, want = ctx.op_rhs(which, F), ctx.op_rhs(which - 2, F), ora.rhs(M, F)
        assert np.array_equal(got, plain), "tiled and plain kernels add the same products in the same order"
        assert rel_fro(got, want) < 1e-14


def _skewed_csc(ora, m, n, seed, sigma=1.3, mean=40.0):
    """columns with log-normal non-zero counts (a few very long ones, many short ones, some empty)"""
    rng = np.random.default_rng(seed)
    want = np.minimum((rng.lognormal(np.log(mean), sigma, n)).astype(np.int64), m)
    want[rng.integers(0, n, 5)] = 0
    xs, is_, p = [], [], [0]
    for c in range(n):
        r = np.sort(rng.choice(m, size=int(want[c]), replace=False))
        is_.append(r)
        xs.append(rng.random(r.size) + 0.25)
        p.append(p[-1] + r.size)
    return ora.CSC(np.concatenate(xs), np.concatenate(is_).astype(np.int32), np.array(p, dtype=np.int32), m, n)


@pytest.mark.parametrize("k", [10, 50, 100])
def test_rhs_tiled_columns_sorted_by_count(sa, ora, k, monkeypatch):
    """The entry stream takes the columns in descending non-zero count (neighbours share a lane pair): on a matrix
    with skewed columns the padding shrinks, and since the order INSIDE a column is untouched the sums are bit for
    bit those of the matrix-order stream (SGL_TILED_SORT=0) and of the plain kernel."""
    monkeypatch.setenv("SGL_TILED_RANGES", "1")   # bit for bit: the tile range whole
    monkeypatch.setenv("SGL_TILED_FULL_TILES", "1")   # the padding figures below are those of LDS-sized tiles
    A = _skewed_csc(ora, 1500, 700, 3)
    At = A.t()
    rng = np.random.default_rng(k)
    W, H = rng.random((A.nrow, k)), rng.random((A.ncol, k))
    out, lay = {}, {}
    for sort in ("1", "0"):
        monkeypatch.setenv("SGL_TILED_SORT", sort)
        c = sa.Context(0)
        try:
            c.upload(to_dgc(sa, A), to_dgc(sa, At))
            out[sort] = (c.op_rhs(2, W), c.op_rhs(3, H), c.op_rhs(0, W), c.op_rhs(1, H))
            c.fit_init(k, ora.synth_winit(k, A.nrow))
            lay[sort] = c.layout_get()
        finally:
            c.close()
    for q in range(2):
        assert np.array_equal(out["1"][q], out["0"][q]) and np.array_equal(out["1"][q], out["1"][q + 2])
    assert rel_fro(out["1"][0], ora.rhs(A, W)) < 1e-14 and rel_fro(out["1"][1], ora.rhs(At, H)) < 1e-14
    nnz = A.p[-1]
    for o in ("A", "At"):
        assert lay["1"][o]["entries"] < lay["0"][o]["entries"]
    assert lay["1"]["A"]["entries"] / nnz < 1.25, lay


@pytest.mark.parametrize("k", [10, 50])
def test_entry_stream_padding_on_pbmc3k(sa, ora, k, monkeypatch):
    """The reference's own data (data/pbmc3k.RData: 13714 genes x 2700 cells, 3 ... 2700 non-zeros per gene): stored
    entries per non-zero of both orientations.  In matrix order the gene side pads 1.76 - 1.81 x (round-2 verdict).  With
    LDS-sized tiles (SGL_TILED_FULL_TILES=1) the sorted stream stores at most 1.20 entries per non-zero; as the fit runs
    it -- a matrix this small gets shorter tiles, so that column groups x tiles reach the 256 CUs -- the padding is the price
    of the parallelism and stays below 1.9."""
    import os
    pads = {}
    for full in ("1", None):
        if full:
            monkeypatch.setenv("SGL_TILED_FULL_TILES", full)
        else:
            monkeypatch.delenv("SGL_TILED_FULL_TILES")
        pads[full] = _pbmc3k_layout(sa, ora, k)
    (lay, nnz), (lay_short, _) = pads["1"], pads[None]
    assert lay["A"]["entries"] / nnz <= 1.20 and lay["At"]["entries"] / nnz <= 1.20, (lay, nnz)
    assert lay_short["A"]["tiles"] > lay["A"]["tiles"] and lay_short["At"]["tiles"] > lay["At"]["tiles"]
    assert lay_short["A"]["entries"] / nnz <= 1.9 and lay_short["At"]["entries"] / nnz <= 1.9, (lay_short, nnz)


def _pbmc3k_layout(sa, ora, k):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pbmc3k_counts.npz"))
    p, dim = g["p"], g["dim"]
    i = g["di"].astype(np.int64)
    for c in range(dim[1]):
        i[p[c]:p[c + 1]] = np.cumsum(i[p[c]:p[c + 1]])
    A = sa.dgCMatrix(g["x"].astype(np.float64), i.astype(np.int32), p, (int(dim[0]), int(dim[1])))
    c = sa.Context(0)
    try:
        c.upload(A, None)
        c.fit_init(k, ora.synth_winit(k, int(dim[0])))
        lay = c.layout_get()
    finally:
        c.close()
    return lay, int(p[-1])


def test_skewed_generator_is_a_valid_consistent_matrix(sa, ora):
    """sgl_synth_csc_skewed (bench.py --data skewed): both orientations describe the same matrix, rows ascend, the
    draw test is the documented one (host emulation with the same hash), and rows / columns ARE skewed."""
    from singlet_amd.context import skew_weights16, SYNTH_SEED
    m, n, inv, off = 700, 520, 10, 3000
    c = sa.Context(0)
    try:
        c.synth(m, n, inv, cell_offset=off, ncells_total=10000, skew=(0.5, 1.5))
        x, i, p = c.download(0)
        xt, it, pt = c.download(1)
    finally:
        c.close()
    A = ora.CSC(x, i, p.astype(np.int32), m, n)
    At = A.t()
    assert np.array_equal(pt, At.p.astype(np.int64)) and np.array_equal(it, At.i) and np.array_equal(xt, At.x)
    for col in range(n):
        assert np.all(np.diff(i[p[col]:p[col + 1]]) > 0)
    cw, gw = skew_weights16(0.5), skew_weights16(1.5)
    S = SYNTH_SEED
    D = np.zeros((m, n), dtype=bool)
    for col in range(0, n, 37):          # a sample of the cells, every gene
        cell = off + col
        lc = (ora.rng_rand(S + 3, cell, 0x5EED) >> 11) & 15
        for g in range(m):
            lg = (ora.rng_rand(S + 4, 0x5EED, g) >> 11) & 15
            u = float(ora.rng_rand(S, cell, g) >> 11) * 2.0 ** -53
            D[g, col] = u < (1.0 / inv) * (cw[lc] * gw[lg])
        got = np.zeros(m, dtype=bool)
        got[i[p[col]:p[col + 1]]] = True
        assert np.array_equal(got, D[:, col]), col
    per_gene = np.diff(pt)
    assert per_gene.max() > 8 * max(np.median(per_gene), 1)     # heavy-tailed gene counts


def test_weight_by_split_one_shot(sa, ora):
    """sgl_c_weight_by_split, the entry `_singlet_weight_by_split` binds (src/singlet.cpp:118-144)."""
    A = ora.synth_csc(300, 257, 10)
    sb = np.random.default_rng(5).integers(0, 4, A.ncol).astype(np.int32)
    ref = ora.weight_by_split(A, sb, 4)
    got = sa.weight_by_split(to_dgc(sa, A), sb, 4)
    assert rel_fro(got.x, ref.x) < 1e-14 and np.array_equal(got.i, A.i) and np.array_equal(got.p, A.p)


def test_rhs_ragged_and_empty_columns(ctx, ora, sa):
    rng = np.random.default_rng(3)
    D = (rng.random((90, 140)) < 0.3) * rng.random((90, 140))
    D[:, 5] = 0
    D[:, 139] = 0
    D[:, 17] = rng.random(90) + 0.1   # full column
    D[40, :] = 0                       # empty row -> empty column of At
    A = ora.CSC(*_csc_from_dense(D))
    ctx.upload(to_dgc(sa, A), to_dgc(sa, A.t()))
    W = rng.random((90, 9))
    B = ctx.op_rhs(0, W)
    assert rel_fro(B, ora.rhs(A, W)) < 1e-14
    assert np.all(B[5] == 0) and np.all(B[139] == 0)


def _csc_from_dense(D):
    nrow, ncol = D.shape
    xs, is_, p = [], [], [0]
    for c in range(ncol):
        r = np.nonzero(D[:, c])[0]
        is_.append(r)
        xs.append(D[r, c])
        p.append(p[-1] + r.size)
    return np.concatenate(xs), np.concatenate(is_), np.array(p), nrow, ncol


# every kernel family of the shared-Gram solve (kernels_nnls.hip dispatch): lane kernel with G as scalar operands (k <= 40),
# DPP rows (42 - 64), x in memory scratch (66 - 104: instances 72, 80, 88, 96, 104), one wave per SIMD (112, 120, 128),
# wave per column above 128 up to SGL_MAX_K
@pytest.mark.parametrize("k", [2, 8, 10, 30, 40, 42, 50, 52, 64, 65, 72, 80, 88, 96, 100, 104, 105, 112, 120, 127, 128, 129, 200, 256, 257, 500])
@pytest.mark.parametrize("L1,L2", [(0.0, 0.0), (0.01, 0.0), (0.01, 0.05)])
def test_nnls(ctx, ora, k, L1, L2):
    rng = np.random.default_rng(k)
    ncols = 300
    F = rng.random((4 * k + 5, k))
    G = ora.aat(F)
    B = rng.normal(size=(ncols, k)) * 3 + 1.0
    X0 = np.abs(rng.normal(size=(ncols, k))) * (rng.random((ncols, k)) < 0.6) * 1e-3
    X, sweeps = ctx.op_nnls(G, B, X0, L1, L2)
    E = np.empty_like(X0)
    esw = 0
    for c in range(ncols):
        E[c], _, it = ora.nnls(G, B[c], X0[c], L1, L2)
        esw += it
    assert rel_fro(X, E) < 1e-10
    assert np.array_equal(X == 0, E == 0)
    assert sweeps == esw


@pytest.mark.parametrize("k", [129, 130, 144, 145, 160, 161, 176, 177, 192, 193, 208, 209, 224, 225, 240, 241, 255, 256])
def test_nnls_ranks_129_to_256_four_columns_per_wave_match_the_oracle_and_the_wave_kernel(ctx, ora, k, monkeypatch):
    """Round 6: ranks 129 - 256 solve four columns per wave (nnls_quad_global_kernel<9 .. 16> against the shared Gram,
    kernels_nnls_quad_big1 / 2.hip) where they fell to the wave-per-column kernel: every instance at both ends of its rank range, a
    ragged column count (a partial quad), skipped columns; the oracle's nnls (src/singlet.cpp:229-250) with equal sweep totals, and
    the bits of the wave kernel."""
    rng = np.random.default_rng(1000 + k)
    ncols = 203
    F = rng.random((3 * k + 5, k))
    G = ora.aat(F)
    B = rng.normal(size=(ncols, k)) * 3 + 1.0
    X0 = np.abs(rng.normal(size=(ncols, k))) * (rng.random((ncols, k)) < 0.5) * 1e-3
    monkeypatch.delenv("SGL_NNLS_NO_QUAD_BIG", raising=False)
    monkeypatch.setenv("SGL_NNLS_NO_QUARTER", "1")
    monkeypatch.setenv("SGL_NNLS_QUAD_GLOBAL_MIN_COLS", "1")      # (short launches above k = 208 keep the wave kernel: count as long)
    X, sweeps = ctx.op_nnls(G, B, X0, 0.01, 0.02)                 # four columns per wave
    esw = 0
    E = np.empty_like(X0)
    for c in range(0, ncols):
        E[c], _, it = ora.nnls(G, B[c], X0[c], 0.01, 0.02)
        esw += it
    assert rel_fro(X, E) < 1e-10 and np.array_equal(X == 0, E == 0) and sweeps == esw
    monkeypatch.setenv("SGL_NNLS_NO_QUAD_BIG", "1")
    Xw, sw = ctx.op_nnls(G, B, X0, 0.01, 0.02)                    # one wave per column
    assert np.array_equal(Xw, X) and sw == sweeps
    # four LANES per column (nnls_quarter.h: the long launches of a plain fit), here forced onto the short launch
    monkeypatch.delenv("SGL_NNLS_NO_QUARTER", raising=False)
    monkeypatch.setenv("SGL_NNLS_QUARTER_MIN_COLS", "1")
    Xq, sq = ctx.op_nnls(G, B, X0, 0.01, 0.02)
    assert np.array_equal(Xq, X) and sq == sweeps


@pytest.mark.parametrize("k", [7, 12, 50, 66, 72, 97, 104, 120])
def test_nnls_repack_passes_are_bit_identical(ctx, k, monkeypatch):
    """The multi-pass lane kernel (stragglers re-packed between passes, nnls_lane.h) must give
    bit-identical solutions and the same sweep total as the one-pass kernel.  Columns with very
    different difficulty (scaled right-hand sides, some warm starts) make the passes uneven."""
    rng = np.random.default_rng(100 + k)
    ncols = 6000
    F = rng.random((4 * k + 5, k))
    G = F.T @ F + 1e-15 * np.eye(k)
    B = rng.normal(size=(ncols, k)) * 3 + 1.0
    B *= np.exp(rng.normal(size=(ncols, 1)) * 2)
    X0 = np.abs(rng.normal(size=(ncols, k))) * (rng.random((ncols, k)) < 0.6) * 1e-3
    monkeypatch.delenv("SGL_NNLS_REPACK_MIN_COLS", raising=False)
    X1, s1 = ctx.op_nnls(G, B, X0, 0.01, 0.0)
    monkeypatch.setenv("SGL_NNLS_REPACK_MIN_COLS", "512")
    X2, s2 = ctx.op_nnls(G, B, X0, 0.01, 0.0)
    assert np.array_equal(X1, X2) and s1 == s2


@pytest.mark.parametrize("k", [65, 71, 72, 73, 88, 96, 97, 100, 104, 105, 111, 112, 113, 120, 121, 127, 128])
def test_nnls_two_lanes_per_column_matches_the_x_scratch_instances(ctx, k, monkeypatch):
    """64 < k <= 128 runs with two lanes per column (nnls_half.h; above 104 with x in AGPRs); SGL_NNLS_NO_HALF=1 selects the lane-per-column
    instances with x in a global scratch.  Same operations in the same order: bit-identical solutions, equal sweep totals,
    one pass or re-packed passes, with a ragged column count (partial waves and workgroups)."""
    rng = np.random.default_rng(300 + k)
    ncols = 5000 + 37
    F = rng.random((3 * k + 5, k))
    G = F.T @ F + 1e-15 * np.eye(k)
    B = rng.normal(size=(ncols, k)) * 3 + 1.0
    B *= np.exp(rng.normal(size=(ncols, 1)) * 2)
    X0 = np.abs(rng.normal(size=(ncols, k))) * (rng.random((ncols, k)) < 0.5) * 1e-3
    out = {}
    for half in (True, False):
        for repack in (False, True):
            if half:
                monkeypatch.delenv("SGL_NNLS_NO_HALF", raising=False)
            else:
                monkeypatch.setenv("SGL_NNLS_NO_HALF", "1")
            if repack:
                monkeypatch.setenv("SGL_NNLS_REPACK_MIN_COLS", "512")
            else:
                monkeypatch.delenv("SGL_NNLS_REPACK_MIN_COLS", raising=False)
            out[(half, repack)] = ctx.op_nnls(G, B, X0, 0.02, 0.01)
    X, s = out[(False, False)]
    for key, (Xo, so) in out.items():
        assert np.array_equal(X, Xo) and s == so, key


def test_scale_and_cor(ctx, ora):
    rng = np.random.default_rng(11)
    F = rng.random((1234, 17))
    S, d = ctx.op_scale(F)
    ES, ed = ora.scale(F)
    assert rel_fro(d, ed) < 1e-14 and rel_fro(S, ES) < 1e-14
    x, y = rng.random(50000), rng.random(50000)
    y = 0.7 * x + 0.3 * y
    assert abs(ctx.op_cor(x, y) - ora.cor(x, y)) < 1e-12
    assert np.isnan(ctx.op_cor(np.ones(10), np.ones(10)))  # zero variance -> NaN ends the loop (quirk 9)


def test_synth_matches_oracle_generator(ctx, ora):
    m, n, off = 257, 130, 1000
    ctx.synth(m, n, 20, cell_offset=off, ncells_total=5000)
    A = ora.synth_csc(m, n, 20, cell0=off)
    x, i, p = ctx.download(0)
    assert np.array_equal(p, A.p.astype(np.int64)) and np.array_equal(i, A.i) and np.array_equal(x, A.x)
    At = A.t()
    x, i, p = ctx.download(1)
    assert np.array_equal(p, At.p.astype(np.int64)) and np.array_equal(i, At.i) and np.array_equal(x, At.x)


def test_device_transpose(ctx, ora, sa):
    A = ora.synth_csc(123, 321, 7)
    ctx.upload(to_dgc(sa, A), None)
    At = A.t()
    x, i, p = ctx.download(1)
    assert np.array_equal(p, At.p.astype(np.int64)) and np.array_equal(i, At.i) and np.array_equal(x, At.x)


def test_log_normalize_and_weight_by_split(sa, ora):
    """Device staging operators against the oracle, on A and on the resident transpose; the float
    tolerance (1e-14 relative) covers the device log1p and the tree-order column sums."""
    A = ora.synth_csc(300, 257, 10)
    counts = ora.CSC(np.round(np.expm1(A.x)) + 1.0, A.i, A.p, A.nrow, A.ncol)
    sb = np.random.default_rng(5).integers(0, 4, A.ncol).astype(np.int32)
    ref_ln = ora.log_normalize(counts, 1e4)
    ref_ws = ora.weight_by_split(ref_ln, sb, 4)
    c = sa.Context(0)
    try:
        c.upload(to_dgc(sa, counts), None)
        c.log_normalize(1e4)
        x, i, p = c.download(0)
        assert np.array_equal(i, counts.i) and np.array_equal(p, counts.p)
        assert rel_fro(x, ref_ln.x) < 1e-14 and np.abs(x - ref_ln.x).max() < 1e-13
        xt, it, pt = c.download(1)
        T = ref_ln.t()
        assert np.array_equal(it, T.i) and np.array_equal(pt, T.p) and rel_fro(xt, T.x) < 1e-14
        c.weight_by_split(sb, 4)
        x2, _, _ = c.download(0)
        assert rel_fro(x2, ref_ws.x) < 1e-14
        xt2, _, _ = c.download(1)
        assert rel_fro(xt2, ref_ws.t().x) < 1e-14
        # the transformed matrix is what the fit then sees
        c.fit_init(5, ora.synth_winit(5, A.nrow))
        c.nmf_run(0.0, 3, 0.01, 0.01, 0.0, 0.0)
        W, d, H = c.get_factors()
        r = ora.c_nmf(ref_ws, ref_ws.t(), 0.0, 3, 0.01, 0.01, 0.0, 0.0, 0, ora.synth_winit(5, A.nrow))
        assert rel_fro(W, r["w"]) < 1e-9 and rel_fro(H, r["h"]) < 1e-9
    finally:
        c.close()
    # R-level mirrors (round trip through the device)
    got = sa.PreprocessData(to_dgc(sa, counts))
    assert rel_fro(got.x, ref_ln.x) < 1e-14
    got2 = sa.weight_by_split(got, sb, 4)
    assert rel_fro(got2.x, ref_ws.x) < 1e-13
    with pytest.raises(sa.SingletHipError):
        sa.weight_by_split(got, sb + 7, 4)   # group ids out of range


@pytest.mark.parametrize("k", [3, 16, 50, 70])
@pytest.mark.parametrize("use_lists", [True, False])
def test_mse_test_op(sa, ora, k, use_lists, monkeypatch):
    """sgl_op_mse_test against ora.mse_test (src/singlet.cpp:536-568) at op level: both kernels (from the cell-side mask
    lists / hashing inside the kernel), one shard and two shards with a cell offset (the hash must see the GLOBAL cell
    index, :590; every shard divides by the total cell count, so the shards' values add up to the whole)."""
    if not use_lists:
        monkeypatch.setenv("SGL_MSE_NO_LIST", "1")
        monkeypatch.setenv("SGL_MASK_NO_LIST", "1")
    m, n, seed, inv = 333, 517, 99, 7
    A = ora.synth_csc(m, n, 9)
    rng = np.random.default_rng(k)
    W = np.abs(rng.standard_normal((m, k)))
    H = np.abs(rng.standard_normal((n, k))) * (rng.random((n, k)) < 0.8)
    d = 0.5 + rng.random(k)
    exp = ora.mse_test(A, W, d, H, seed, inv)

    def shard(lo, hi):
        sub = ora.CSC(A.x[A.p[lo]:A.p[hi]], A.i[A.p[lo]:A.p[hi]], A.p[lo:hi + 1] - A.p[lo], m, hi - lo)
        c = sa.Context(0)
        try:
            c.upload(to_dgc(sa, sub), None, cell_offset=lo, ncells_total=n)
            c.fit_init(k, W)
            if use_lists:   # the lists are built by the masked H-update of a fit; mse_test alone must build what it needs
                pass
            c.set_factors(W, d, H[lo:hi])
            return c.op_mse_test(seed, inv)
        finally:
            c.close()

    one = shard(0, n)
    assert abs(one - exp) <= 1e-11 * abs(exp), (one, exp)
    two = shard(0, 200) + shard(200, n)
    assert abs(two - exp) <= 1e-11 * abs(exp), (two, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [7, 50, 100])
def test_mse_test_from_listed_values_equals_the_window_kernel(sa, ora, k, monkeypatch):
    """Once a masked H-update has listed the mask, mse_test reads the matrix values at the listed entries from a list
    made on its first call (mask_vals_kernel) instead of searching the cell's non-zeros in every trace: the bits of the
    window kernel (SGL_MSE_NO_VALS=1), the oracle's value (src/singlet.cpp:536-568), also with a cell offset, empty
    cells and a cell whose every gene is a non-zero."""
    m, n, seed, inv = 333, 517, 99, 7
    A = ora.synth_csc(m, n, 9)
    # an empty cell and a full one
    p = A.p.copy()
    cnt = np.diff(p)
    x2, i2, p2 = [], [], [0]
    for c in range(n):
        if c == 5:
            pass
        elif c == 11:
            x2.append(1.0 + np.arange(m) % 5); i2.append(np.arange(m, dtype=np.int32))
        else:
            x2.append(A.x[p[c]:p[c + 1]]); i2.append(A.i[p[c]:p[c + 1]])
        p2.append(p2[-1] + (0 if c == 5 else (m if c == 11 else cnt[c])))
    A = ora.CSC(np.concatenate(x2).astype(np.float64), np.concatenate(i2).astype(np.int32), np.asarray(p2, dtype=np.int64), m, n)
    rng = np.random.default_rng(k)
    W = np.abs(rng.standard_normal((m, k)))
    H = np.abs(rng.standard_normal((n, k))) * (rng.random((n, k)) < 0.8)
    d = 0.5 + rng.random(k)
    exp = ora.mse_test(A, W, d, H, seed, inv)

    def shard(lo, hi):
        sub = ora.CSC(A.x[A.p[lo]:A.p[hi]], A.i[A.p[lo]:A.p[hi]], A.p[lo:hi + 1] - A.p[lo], m, hi - lo)
        c = sa.Context(0)
        try:
            c.upload(to_dgc(sa, sub), None, cell_offset=lo, ncells_total=n)
            c.fit_init(k, W)
            c.step_h_masked(0.0, 0.0, seed, inv)          # lists the mask of the cell side
            assert c.layout_builds()[2] == 1
            c.set_factors(W, d, H[lo:hi])
            monkeypatch.setenv("SGL_MSE_NO_VALS", "1")
            window = c.op_mse_test(seed, inv)
            monkeypatch.delenv("SGL_MSE_NO_VALS")
            first = c.op_mse_test(seed, inv)               # makes the value list
            again = c.op_mse_test(seed, inv)               # reads it
            return window, first, again
        finally:
            c.close()

    w1, f1, a1 = shard(0, n)
    assert f1 == w1 and a1 == w1
    assert abs(f1 - exp) <= 1e-11 * abs(exp), (f1, exp)
    parts = [shard(0, 200), shard(200, n)]
    assert all(f == w and a == w for w, f, a in parts)
    assert abs(sum(q[1] for q in parts) - exp) <= 1e-11 * abs(exp)


def _random_csc(ora, rng, m, n, style):
    """random sparse matrices with awkward columns: empty ones, single entries, dense runs, heavy tails"""
    if style == "uniform":
        cnt = rng.binomial(m, min(1.0, rng.uniform(0.002, 0.3)), n)
    elif style == "heavy":
        cnt = np.minimum((rng.lognormal(np.log(max(m * 0.02, 1.0)), 1.5, n)).astype(np.int64), m)
    elif style == "mostly_empty":
        cnt = np.where(rng.random(n) < 0.85, 0, rng.integers(1, max(2, m // 3), n))
    else:   # "extremes": empty, one entry, full columns side by side
        cnt = rng.choice([0, 1, 2, m // 2, m], size=n)
    xs, is_, p = [], [], [0]
    for c in range(n):
        r = np.sort(rng.choice(m, size=int(min(cnt[c], m)), replace=False))
        is_.append(r)
        xs.append(rng.random(r.size) + 0.25)
        p.append(p[-1] + r.size)
    return ora.CSC(np.concatenate(xs) if xs else np.zeros(0), (np.concatenate(is_) if is_ else np.zeros(0)).astype(np.int32),
                   np.array(p, dtype=np.int32), m, n)


@pytest.mark.parametrize("case", range(96))
def test_rhs_tiled_random_structures(sa, ora, case, monkeypatch):
    """Randomised shapes against the plain kernel and the oracle: ranks on both sides of every layout boundary (16 / 17, 32 / 33,
    64 / 65, 128), row counts around the tile sizes (408, 632, 984 rows), column counts around the wave-block sizes (64, 128,
    512, 1024), empty / single-entry / full columns, whole range and forced tile-range splits.  Whole range: bit-equal to the
    plain kernel (same products in the same order); split: equal to rounding."""
    rng = np.random.default_rng(1000 + case)
    k = int(rng.choice([1, 2, 3, 8, 15, 16, 17, 24, 31, 32, 33, 40, 50, 63, 64, 65, 96, 128]))
    m = int(rng.choice([1, 7, 63, 407, 408, 409, 631, 632, 633, 983, 985, 1300, 2100]))
    n = int(rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 511, 513, 1025, 1500]))
    style = ["uniform", "heavy", "mostly_empty", "extremes"][case % 4]
    ranges = int(rng.choice([0, 1, 1, 2, 3]))   # 0: the library's own choice (these matrices are small: usually a split)
    if ranges:
        monkeypatch.setenv("SGL_TILED_RANGES", str(ranges))
    if case % 3 == 0:
        monkeypatch.setenv("SGL_TILED_FULL_TILES", "1")   # LDS-sized tiles (row counts around the tile sizes); otherwise the short tiles of a small matrix
    A = _random_csc(ora, rng, m, n, style)
    if A.nnz == 0:
        A = _random_csc(ora, rng, m, n, "uniform")
    At = A.t()
    W, H = rng.random((m, k)), rng.random((n, k))
    c = sa.Context(0)
    try:
        c.upload(to_dgc(sa, A), to_dgc(sa, At))
        res = [(c.op_rhs(2, W), c.op_rhs(0, W), ora.rhs(A, W)), (c.op_rhs(3, H), c.op_rhs(1, H), ora.rhs(At, H))]
    finally:
        c.close()
    for got, plain, want in res:
        assert np.all(np.isfinite(got))
        assert rel_fro(got, want) < 1e-13, (k, m, n, style, ranges)
        if k <= 64 and ranges == 1:
            assert np.array_equal(got, plain), (k, m, n, style, ranges)
        else:
            assert rel_fro(got, plain) < 1e-13


@pytest.mark.parametrize("k", [1, 7, 12, 16, 17, 30, 32, 33, 48, 49, 50, 63, 64])
def test_nnls_four_columns_per_wave_on_a_shared_gram_matches_the_lane_kernel(ctx, ora, k, monkeypatch):
    """Short launches against a shared Gram (a rank's gene block on a team, small matrices) run four columns per wave with the
    Gram staged in LDS (nnls_quad_shared_kernel, round 5) instead of a lane per column: same operations in the same order --
    bit-identical solutions and equal sweep totals, ragged column counts (partial quads, partial workgroups), and the oracle's
    solution to 1e-9."""
    rng = np.random.default_rng(700 + k)
    F = rng.random((3 * k + 5, k))
    G = F.T @ F + 1e-15 * np.eye(k)
    for ncols in (1, 3, 4, 1000 + 37, 3750):
        B = rng.normal(size=(ncols, k)) * 3 + 1.0
        B *= np.exp(rng.normal(size=(ncols, 1)) * 2)
        X0 = np.abs(rng.normal(size=(ncols, k))) * (rng.random((ncols, k)) < 0.5) * 1e-3
        monkeypatch.delenv("SGL_OP_NNLS_QUAD_SHARED", raising=False)
        Xl, sl = ctx.op_nnls(G, B, X0, 0.02, 0.01)
        monkeypatch.setenv("SGL_OP_NNLS_QUAD_SHARED", "1")
        Xq, sq = ctx.op_nnls(G, B, X0, 0.02, 0.01)
        assert np.array_equal(Xl, Xq) and sl == sq, (k, ncols)
    tot = 0
    for c in range(40):     # the oracle solves one column per call
        xo, _, it = ora.nnls(G, B[c], X0[c], 0.02, 0.01)
        tot += it
        assert np.linalg.norm(Xq[c] - xo) <= 1e-9 * max(np.linalg.norm(xo), 1e-300) and np.array_equal(Xq[c] == 0, xo == 0)
    Xs, ss = ctx.op_nnls(G, B[:40], X0[:40], 0.02, 0.01)
    assert ss == tot and np.array_equal(Xs, Xq[:40])


HALF_ASM_RANKS = [65, 66, 68, 69, 72, 75, 76, 79, 80, 83, 84, 88, 89, 92, 95, 96, 97, 98, 99, 100, 101, 104, 105, 108, 111, 112, 116, 117, 120, 123, 124, 125, 127, 128]


@pytest.mark.parametrize("k", HALF_ASM_RANKS)
def test_nnls_generated_two_lane_solve_matches_the_compiled_kernel(ctx, ora, k, monkeypatch):
    """Round 5: ranks 65 ... 128 against a shared Gram run the two-lanes-per-column solve as generated asm (gen_nnls_half.py: one
    half computes a coordinate's step under a gate, only nd crosses the halves, the step's chain interleaved with the row
    updates; padded to multiples of 4 where the compiled instances pad to 8; above 100 with x in the accumulator registers).  Bit-identical solutions and equal sweep totals
    against the hipcc-scheduled kernel (SGL_NNLS_NO_ASM=1): one pass and re-packed passes, ragged column counts down to one,
    a launch long enough for the 512-thread workgroups, penalties on and off, warm starts; and the oracle column by column."""
    rng = np.random.default_rng(1900 + k)
    F = rng.random((3 * k + 5, k))
    G = F.T @ F + 1e-15 * np.eye(k)
    for ncols, L1, L2 in ((1, 0.0, 0.0), (31, 0.02, 0.0), (32 * 5 + 7, 0.02, 0.01), (3000 + 37, 0.0, 0.03), (66000 + 5, 0.01, 0.0)):
        if ncols > 60000 and k not in (72, 84, 100, 116, 128):
            continue
        B = rng.normal(size=(ncols, k)) * 3 + 1.0
        B *= np.exp(rng.normal(size=(ncols, 1)) * 2)
        X0 = np.abs(rng.normal(size=(ncols, k))) * (rng.random((ncols, k)) < 0.5) * 1e-3
        out = {}
        for asm in (True, False):
            for repack in (False, True):
                if asm:
                    monkeypatch.delenv("SGL_NNLS_NO_ASM", raising=False)
                else:
                    monkeypatch.setenv("SGL_NNLS_NO_ASM", "1")
                if repack:
                    monkeypatch.setenv("SGL_NNLS_REPACK_MIN_COLS", "512")
                else:
                    monkeypatch.delenv("SGL_NNLS_REPACK_MIN_COLS", raising=False)
                out[(asm, repack)] = ctx.op_nnls(G, B, X0, L1, L2)
        X, s = out[(False, False)]
        for key, (Xo, so) in out.items():
            assert np.array_equal(X, Xo) and s == so, (key, ncols)
        if ncols == 3037:
            for c in range(12):
                xo, _, it = ora.nnls(G, B[c], X0[c], L1, L2)
                assert np.linalg.norm(X[c] - xo) <= 1e-9 * max(np.linalg.norm(xo), 1e-300) and np.array_equal(X[c] == 0, xo == 0)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 7, 9, 10, 15, 16, 17, 19, 20, 24, 29, 30, 31, 32, 33, 39, 40, 41, 43, 44, 47, 48, 49, 50, 51, 52, 55, 56, 59, 60, 63, 64])
def test_nnls_generated_sweep_matches_the_compiled_kernel(ctx, ora, k, monkeypatch):
    """Round 5: at the ranks it has instances for, the lane-per-column solve runs its sweep as generated, hand-scheduled asm
    (gen_nnls_lane.py: a coordinate's serial step chain interleaved with its neighbours' row-update FMAs).  Same operations in
    the same order per column: bit-identical solutions and equal sweep totals against the hipcc-scheduled kernel
    (SGL_NNLS_NO_ASM=1), one pass and re-packed passes, ragged column counts, penalties on and off, warm and cold starts; and
    the oracle's solution column by column."""
    rng = np.random.default_rng(900 + k)
    F = rng.random((3 * k + 5, k))
    G = F.T @ F + 1e-15 * np.eye(k)
    for ncols, L1, L2 in ((1, 0.0, 0.0), (63, 0.02, 0.0), (64 * 5 + 7, 0.02, 0.01), (5000 + 37, 0.0, 0.03)):
        B = rng.normal(size=(ncols, k)) * 3 + 1.0
        B *= np.exp(rng.normal(size=(ncols, 1)) * 2)
        X0 = np.abs(rng.normal(size=(ncols, k))) * (rng.random((ncols, k)) < 0.5) * 1e-3
        out = {}
        for asm in (True, False):
            for repack in (False, True):
                if asm:
                    monkeypatch.delenv("SGL_NNLS_NO_ASM", raising=False)
                else:
                    monkeypatch.setenv("SGL_NNLS_NO_ASM", "1")
                if repack:
                    monkeypatch.setenv("SGL_NNLS_REPACK_MIN_COLS", "512")
                else:
                    monkeypatch.delenv("SGL_NNLS_REPACK_MIN_COLS", raising=False)
                out[(asm, repack)] = ctx.op_nnls(G, B, X0, L1, L2)
        X, s = out[(False, False)]
        for key, (Xo, so) in out.items():
            assert np.array_equal(X, Xo) and s == so, (key, ncols)
    for c in range(30):
        xo, _, it = ora.nnls(G, B[c], X0[c], L1, L2)
        assert np.linalg.norm(X[c] - xo) <= 1e-9 * max(np.linalg.norm(xo), 1e-300) and np.array_equal(X[c] == 0, xo == 0)
