"""BASELINE config 4 -- 30 000 genes x 1 000 000 cells, k = 50, the cells SHARDED over the ranks of a team -- under the oracle on
one device.  An 8-GPU node is not available to the test box, so the eight ranks (and a seven-rank team, whose gene blocks do not
divide the genes: the last block is padded) share device 0: the whole team logic runs as it would across devices -- every rank's
shard generated at its cell offset (875 000 for the last of eight), its own entry streams and mask lists, gene blocks of 3750,
the reduce-scatter / all-reduce / all-gather steps with their k * mb units -- only the transport is a summing HIP kernel instead
of RCCL (which refuses duplicate devices).

The columns of predict / predict_mask are independent (src/singlet.cpp:339-346, :445-465), so the oracle follows any slice of
the full problem exactly, as in tests/test_gpu_fullsize_oracle.py: H slices STRADDLING EVERY RANK BOUNDARY (the reference's own
idiom for a column block inside a larger matrix is the chunk loop with a running offset, :384-402, `i + offset` in the mask hash
:485, `j + offset` :590), whole gene columns of w at the edges of the gene blocks, each a sum over all 10^6 cells of all ranks."""
import numpy as np
import pytest

from conftest import rel_fro, same_zero_pattern

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(2400)]

GENES, CELLS, INV, K = 30000, 1000000, 20, 50
L1 = 0.01
SEED, INV_MASK = 4711, 20


def _cell_lo(ranks):
    base, rem = divmod(CELLS, ranks)
    lo = [0]
    for r in range(ranks):
        lo.append(lo[-1] + base + (1 if r < rem else 0))
    return lo


def _h_slices(ranks, half):
    """(first cell, width): the first cells, `half` cells either side of EVERY rank boundary, the last cells"""
    lo = _cell_lo(ranks)
    return [(0, 2 * half)] + [(b - half, 2 * half) for b in lo[1:-1]] + [(CELLS - 2 * half, 2 * half)]


def _gene_groups(ranks, cnt):
    mb = (GENES + ranks - 1) // ranks                     # genes per rank block (multi.hip: team_iterate)
    last0 = (ranks - 1) * mb                              # first gene of the last (for 7 ranks: short, padded) block
    heavy, light = int(np.argmax(cnt)), int(np.argmin(cnt))
    return [[0, 1, 2], [mb - 1, mb], [last0 - 1, last0], [GENES - 2, GENES - 1], [heavy], [light]]


@pytest.fixture(scope="module")
def one(sa, ora):
    """The same two iterations on ONE context (itself held against the oracle by slices in test_gpu_fullsize_oracle.py): the
    figures a slice cannot give -- the row sums of the unscaled h over all 10^6 cells, tol, the test error."""
    c = sa.Context(0)
    try:
        c.synth(GENES, CELLS, INV)
        cnt = c.col_counts(1)
        out = {"gene_counts": cnt}
        for name in ("plain", "masked"):
            c.fit_init(K, None)
            c.step_begin()
            if name == "plain":
                c.step_h(L1, 0.0)
            else:
                c.step_h_masked(L1, 0.0, SEED, INV_MASK)
            c.step_scale_h()
            _, dh, Hs = c.get_factors(w=False)
            if name == "plain":
                c.step_w(L1, 0.0)
            else:
                c.step_w_masked(L1, 0.0, SEED, INV_MASK)
            tol = c.step_scale_w()
            W, dw, _ = c.get_factors(h=False)
            out[name] = dict(dh=dh, Hs=Hs, tol=tol, W=W, d=dw)
        out["masked"]["mse"] = c.op_mse_test(SEED, INV_MASK)
    finally:
        c.close()
    return out


@pytest.fixture(scope="module", params=[8, 7])
def team(request, sa):
    with sa.Multi([0] * request.param) as M:
        M.synth(GENES, CELLS, INV)
        M.ranks = request.param
        yield M


def _replicas_agree(M, W, d):
    for r in range(1, M.ranks):
        Wr, dr, _ = M.rank_ctx(r).get_factors(h=False)
        assert np.array_equal(Wr, W) and np.array_equal(dr, d), r


def test_config4_plain_iteration_on_a_team_equals_the_oracle(one, team, ora):   # `one` first: it is gone before the team is made
    """One c_nmf iteration (:650-659) at k = 50 with the cells sharded over 8 / 7 ranks."""
    M, ref1 = team, one["plain"]
    M.fit_init(K, None)
    W0 = ora.synth_winit(K, GENES)
    tol = M.iterate(L1, L1, 0.0, 0.0)
    W, d, H = M.get_factors()                                     # w, h scaled (the state after :659), d = row sums of w
    _replicas_agree(M, W, d)
    # h: every slice around a rank boundary is the oracle's predict on the regenerated cells, scaled by the GLOBAL row sums
    for s0, width in _h_slices(M.ranks, 256):
        A_s = ora.synth_csc(GENES, width, INV, cell0=s0)
        ref = ora.predict(A_s, W0, np.zeros((width, K)), L1, 0.0) / ref1["dh"]
        got = H[s0:s0 + width]
        assert rel_fro(got, ref) < 1e-9, (s0, rel_fro(got, ref))
        assert same_zero_pattern(got, ref), s0
    assert np.abs(H.sum(axis=0) - 1.0).max() < 1e-9                # scale(h, d) over ALL ranks' cells
    assert rel_fro(H, ref1["Hs"]) < 1e-11                          # every cell of every rank: the one-context fit to rounding
    # w: genes at the gene-block edges (reduce-scatter units of k * mb doubles; 7 ranks: mb * N > m, the last block padded),
    # the first / last / heaviest / lightest gene -- each b_g and the Gram are sums over all ranks' cells
    for genes in _gene_groups(M.ranks, one["gene_counts"]):
        G = ora.synth_gene_columns(genes, CELLS, INV)
        ref = ora.predict(G, H, W0[genes].copy(), L1, 0.0)
        got = W[genes] * d                                         # unscaled w
        assert rel_fro(got, ref) < 1e-9, (genes, rel_fro(got, ref))
        assert same_zero_pattern(got, ref), genes
    assert rel_fro(W, ref1["W"]) < 1e-10 and rel_fro(d, ref1["d"]) < 1e-10
    assert abs(tol - ref1["tol"]) <= 1e-9 * abs(ref1["tol"]), (tol, ref1["tol"])


def test_config4_masked_iteration_on_a_team_equals_the_oracle(one, team, ora):
    """One c_ard_nmf iteration (:1108-1111) + its trace row on the team: the mask hash sees the GLOBAL cell index on every rank
    (`i + offset`, :485; `j + offset` in mse_test, :590), the per-gene Gram downdates (k x k x genes: 600 MB per rank) are
    reduce-scattered by gene blocks."""
    M, ref1 = team, one["masked"]
    M.fit_init(K, None)
    W0 = ora.synth_winit(K, GENES)
    r = M.ard_run(0.0, 1, L1, 0.0, SEED, INV_MASK, 1e9, 1)
    assert list(r["iter"]) == [0] and r["n_iter"] == 1
    W, d, H = M.get_factors()
    _replicas_agree(M, W, d)
    for s0, width in _h_slices(M.ranks, 64):
        A_s = ora.synth_csc(GENES, width, INV, cell0=s0)
        ref = ora.predict_mask(A_s, SEED, INV_MASK, W0, np.zeros((width, K)), L1, 0.0, col_offset=s0) / ref1["dh"]
        got = H[s0:s0 + width]
        assert rel_fro(got, ref) < 1e-9, (s0, rel_fro(got, ref))
        assert same_zero_pattern(got, ref), s0
    assert rel_fro(H, ref1["Hs"]) < 1e-11
    for genes in _gene_groups(M.ranks, one["gene_counts"])[:5]:
        G = ora.synth_gene_columns(genes, CELLS, INV)
        ref = ora.predict_mask(G, SEED, INV_MASK, H, W0[genes].copy(), L1, 0.0, mask_t=True, col_offset=genes[0])
        got = W[genes] * d
        assert rel_fro(got, ref) < 1e-9, (genes, rel_fro(got, ref))
        assert same_zero_pattern(got, ref), genes
    assert rel_fro(W, ref1["W"]) < 1e-10 and rel_fro(d, ref1["d"]) < 1e-10
    assert abs(r["tol"][0] - ref1["tol"]) <= 1e-9 * abs(ref1["tol"])
    assert abs(r["test_mse"][0] - ref1["mse"]) <= 1e-10 * abs(ref1["mse"]), (r["test_mse"][0], ref1["mse"])
