"""CPU stand-in for singlet_amd.Context used ONLY by the gloo tests of the sharded host
logic: the same step API, computed with the oracle.  Test infrastructure, not product."""
import numpy as np


class OracleShardContext:
    def __init__(self, ora):
        self.ora = ora
        self.allreduce = None

    def upload(self, A, At, cell_offset=0, ncells_total=0):
        self.A, self.At = A, At
        self.cell_offset, self.ncells_total = cell_offset, ncells_total or A.ncol
        self.gene_nnz = np.diff(At.p).astype(np.float64)

    def set_allreduce(self, fn):
        self.allreduce = fn

    def _sum(self, arr):
        if self.allreduce is not None:
            self.allreduce(arr)

    def fit_init(self, k, w_init):
        self.k = k
        self.W = np.array(w_init, dtype=np.float64, order="C")
        self.H = np.zeros((self.A.ncol, k))
        self.d = np.ones(k)
        g = self.gene_nnz.copy()
        self._sum(g)               # global per-gene counts decide which W columns are skipped (l.340)
        self.gene_nnz_global = g

    def step_begin(self):
        self.Wprev = self.W.copy()

    def step_h(self, L1, L2):
        self.H = self.ora.predict(self.A, self.W, self.H, L1, L2)

    def step_scale_h(self):
        d = np.zeros(self.k)
        for c in range(self.H.shape[0]):
            d += self.H[c]
        self._sum(d)
        d += 1e-15
        self.H = self.H / d[None, :]
        self.d = d

    def step_w(self, L1, L2):
        k, m = self.k, self.A.nrow
        red = np.empty(k * m + k * k)
        red[:k * m] = self.ora.rhs(self.At, self.H).ravel()
        red[k * m:] = (self.ora.aat(self.H) - 1e-15 * np.eye(k)).ravel() if self.H.shape[0] else 0.0
        self._sum(red)
        Bw = red[:k * m].reshape(m, k)
        G = red[k * m:].reshape(k, k).copy()
        G[np.diag_indices(k)] += 1e-15
        for g in range(m):
            if self.gene_nnz_global[g] == 0:
                continue
            self.W[g], _, _ = self.ora.nnls(G, Bw[g], self.W[g], L1, L2)

    def step_scale_w(self):
        self.W, self.d = self.ora.scale(self.W)
        return self.ora.cor(self.W, self.Wprev)

    def get_factors(self):
        return self.W, self.d, self.H


class OracleTeamRank:
    """CPU stand-in for one rank of the library's NATIVE team (singlet_amd/csrc/multi.hip, team_iterate):
    the same exchange pattern -- partials of the UNSCALED h, one grouped exchange (reduce-scatter of the
    k x genes right-hand sides by gene blocks + all-reduce of [Gram | row sums]), the rank's block of genes
    solved, all-gather of the w blocks, replicated scale / cor -- computed with the oracle's operators and
    exchanged through the three collectives it is handed (gloo in tests/test_sharded_gloo.py)."""

    def __init__(self, ora, rank, world, reduce_scatter, all_reduce, all_gather):
        self.ora, self.rank, self.world = ora, rank, world
        self.reduce_scatter, self.all_reduce, self.all_gather = reduce_scatter, all_reduce, all_gather

    def upload(self, A, At):
        self.A, self.At = A, At
        g = np.diff(At.p).astype(np.float64)
        self.all_reduce(g)                 # global per-gene counts (src/singlet.cpp:340 skip rule)
        self.gene_nnz_global = g

    def fit_init(self, k, w_init):
        self.k = k
        m = self.A.nrow
        self.mb = (m + self.world - 1) // self.world
        self.W = np.zeros((self.mb * self.world, k))   # padded to equal gene blocks, as the device buffers are
        self.W[:m] = w_init
        self.H = np.zeros((self.A.ncol, k))
        self.d = np.ones(k)

    def iterate(self, L1_w, L1_h, L2_w, L2_h):
        ora, k, m, mb, r = self.ora, self.k, self.A.nrow, self.mb, self.rank
        Wprev = self.W[:m].copy()
        self.H = ora.predict(self.A, np.ascontiguousarray(self.W[:m]), self.H, L1_h, L2_h)
        Bw = np.zeros((mb * self.world, k))
        Bw[:m] = ora.rhs(self.At, self.H)                                     # unscaled partial right-hand sides
        tail = np.empty(k * k + k)
        tail[:k * k] = (ora.aat(self.H) - 1e-15 * np.eye(k)).ravel() if self.H.shape[0] else 0.0
        tail[k * k:] = self.H.sum(axis=0) if self.H.shape[0] else 0.0
        blk = self.reduce_scatter(Bw)                                         # this rank's gene block, summed
        self.all_reduce(tail)
        d = tail[k * k:] + 1e-15
        self.H = self.H / d[None, :]                                          # scale(h, d) with the global sums
        G = tail[:k * k].reshape(k, k) / d[:, None] / d[None, :]
        G[np.diag_indices(k)] += 1e-15
        g0 = r * mb
        Wb = self.W[g0:g0 + mb].copy()
        for q in range(max(0, min(mb, m - g0))):
            if self.gene_nnz_global[g0 + q] == 0:
                continue
            Wb[q], _, _ = ora.nnls(G, blk[q] / d, Wb[q], L1_w, L2_w)
        self.W = self.all_gather(Wb)
        Ws, self.d = ora.scale(np.ascontiguousarray(self.W[:m]))
        self.W[:m] = Ws
        return ora.cor(Ws, Wprev)

    def get_factors(self):
        return self.W[:self.A.nrow], self.d, self.H
