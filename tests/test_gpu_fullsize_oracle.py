"""Oracle parity AT THE SIZE THE BENCH RUNS (BASELINE configs 3 and 5: 30 000 genes x 1 000 000 cells, 1.5e9
non-zeros), by column slices.  The columns of predict / predict_mask are independent (src/singlet.cpp:339-346,
:445-465), so the oracle can follow any slice of them exactly: the slice's cells are regenerated on the host
(ora.synth_csc(cell0=...)), a few whole gene columns of t(A) likewise (ora.synth_gene_columns), and the oracle
solves them as the full matrix would -- with the GLOBAL cell index in the mask hash (the reference's `i + offset`,
:485).  This is where the entry-stream offsets above 2^32 bytes, the last column blocks, all 74 row tiles on the
H side and the 2451 row tiles / 13+ tile ranges on the W side meet the oracle, not only the repo's own plain kernel."""
import numpy as np
import pytest

from conftest import rel_fro, same_zero_pattern

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1800)]

GENES, CELLS, INV = 30000, 1000000, 20
L1 = 0.01
SEED, INV_MASK = 4711, 20


@pytest.fixture(scope="module")
def full(sa):
    c = sa.Context(0)
    c.synth(GENES, CELLS, INV)
    yield c
    c.close()


def _cell_slices(width):
    # first cells, around the middle (not aligned to a 64-column wave block), the very last ones (stream offsets > 2^32 B)
    return [0, 499744 + 17, CELLS - width]


def _gene_picks(ctx):
    cnt = ctx.col_counts(1)
    assert cnt.shape == (GENES,) and int(cnt.sum()) == ctx.dims()[2]
    heavy, light = int(np.argmax(cnt)), int(np.argmin(cnt))
    groups = [[0, 1, 2], [GENES - 2, GENES - 1], [heavy], [light]]
    return groups, cnt


@pytest.mark.parametrize("k", [50, 20, 160])
def test_config3_h_and_w_update_slices_equal_the_oracle(full, ora, k):
    """One H-update and one W-update of c_nmf at k = 50 (and k = 20: the four-columns-per-LDS-instruction stream with its
    tile-range split at this size; k = 160, round 6: five quad passes over the stream, the Gram of ranks 129 - 256 on the matrix
    cores, the four-lanes-per-column solve on 10^6 columns) on the resident config-3 matrix: 3 x 512 cells of h and 7 genes of w
    (first, last, heaviest, lightest; at k = 160 the first three only -- the oracle's AAt(h) per call is serial) against ora.predict
    on the regenerated slices."""
    width = 512 if k <= 64 else 128
    full.fit_init(k, None)
    W0 = ora.synth_winit(k, GENES)
    Wdev, _, _ = full.get_factors(h=False)
    assert np.array_equal(Wdev, W0)                                   # same start as the oracle's generator
    full.step_begin()
    full.step_h(L1, 0.0)                                              # h = predict(A, w, h = 0)  (:650)
    _, _, H = full.get_factors(w=False, d=False)
    for s0 in _cell_slices(width):
        A_s = ora.synth_csc(GENES, width, INV, cell0=s0)
        ref = ora.predict(A_s, W0, np.zeros((width, k)), L1, 0.0)
        got = H[s0:s0 + width]
        assert rel_fro(got, ref) < 1e-9, (s0, rel_fro(got, ref))
        assert same_zero_pattern(got, ref), s0
    # columns outside the slices are sane too
    assert np.all(np.isfinite(H)) and np.all(H >= 0)
    full.step_scale_h()                                               # scale(h, d)  (:651)
    _, dh, Hs = full.get_factors(w=False)
    assert np.abs(Hs.sum(axis=0) - 1.0).max() < 1e-9
    # d = rowsums over all 10^6 cells (+ 1e-15), h /= d: against the host's sums of the unscaled h (:219-225)
    d_host = H.sum(axis=0) + 1e-15
    assert rel_fro(dh, d_host) < 1e-12
    assert rel_fro(Hs[:4096], H[:4096] / d_host) < 1e-12 and rel_fro(Hs[-4096:], H[-4096:] / d_host) < 1e-12
    full.step_w(L1, 0.0)                                              # w = predict(At, h, w)  (:654), warm start w0
    W1, _, _ = full.get_factors(h=False)
    groups, cnt = _gene_picks(full)
    for genes in (groups if k <= 64 else groups[:1]):
        G = ora.synth_gene_columns(genes, CELLS, INV)
        assert np.array_equal(np.diff(G.p), cnt[genes])               # the device's t(A) holds exactly these columns
        ref = ora.predict(G, Hs, W0[genes].copy(), L1, 0.0)           # a = AAt(h) over all 1e6 cells inside
        got = W1[genes]
        assert rel_fro(got, ref) < 1e-9, (genes, rel_fro(got, ref))
        assert same_zero_pattern(got, ref), genes
    # scale(w, d); tol = cor(w, w_it)  (:655-659): the oracle's scale and one-pass cor on the downloaded, unscaled w
    tol = full.step_scale_w()
    W2, dw, _ = full.get_factors(h=False)
    Ws, d_ref = ora.scale(W1)
    assert rel_fro(dw, d_ref) < 1e-12 and rel_fro(W2, Ws) < 1e-12
    tol_ref = ora.cor(Ws, W0)
    assert abs(tol - tol_ref) <= 1e-8 * abs(tol_ref), (tol, tol_ref)


@pytest.mark.parametrize("k,width", [(50, 256), (100, 128), (10, 256)])
def test_config5_masked_h_and_w_update_slices_equal_the_oracle(full, ora, k, width):
    """The masked half-iterations of c_ard_nmf (predict_mask, :436-466) at k = 50 and k = 100 on the config-5 matrix:
    slices of h with the global cell index in the hash, whole gene columns of w (mask_t = true: draw(cell, gene))."""
    full.fit_init(k, None)
    W0 = ora.synth_winit(k, GENES)
    full.step_begin()
    full.step_h_masked(L1, 0.0, SEED, INV_MASK)
    _, _, H = full.get_factors(w=False, d=False)
    for s0 in _cell_slices(width):
        A_s = ora.synth_csc(GENES, width, INV, cell0=s0)
        ref = ora.predict_mask(A_s, SEED, INV_MASK, W0, np.zeros((width, k)), L1, 0.0, col_offset=s0)
        got = H[s0:s0 + width]
        assert rel_fro(got, ref) < 1e-9, (k, s0, rel_fro(got, ref))
        assert same_zero_pattern(got, ref), (k, s0)
    full.step_scale_h()
    _, _, Hs = full.get_factors(w=False, d=False)
    full.step_w_masked(L1, 0.0, SEED, INV_MASK)
    W1, _, _ = full.get_factors(h=False)
    groups, cnt = _gene_picks(full)
    for genes in groups[:3]:                                          # first, last, heaviest (the oracle's AAt(h) per call is serial)
        G = ora.synth_gene_columns(genes, CELLS, INV)
        ref = ora.predict_mask(G, SEED, INV_MASK, Hs, W0[genes].copy(), L1, 0.0, mask_t=True, col_offset=genes[0])
        got = W1[genes]
        assert rel_fro(got, ref) < 1e-9, (k, genes, rel_fro(got, ref))
        assert same_zero_pattern(got, ref), (k, genes)
