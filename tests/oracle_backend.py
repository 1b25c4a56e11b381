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
