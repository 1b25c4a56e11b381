"""dgCMatrix slots as the reference's Rcpp::SparseMatrix sees them
(inst/include/singlet.h:36-44): x (double), i (int32 row index, ascending
within a column), p (int32 column pointers), Dim = (nrow, ncol)."""
import numpy as np


class dgCMatrix:
    __slots__ = ("x", "i", "p", "Dim", "Dimnames")

    def __init__(self, x, i, p, Dim, Dimnames=(None, None)):
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.i = np.ascontiguousarray(i, dtype=np.int32)
        self.p = np.ascontiguousarray(p, dtype=np.int32)
        self.Dim = (int(Dim[0]), int(Dim[1]))
        self.Dimnames = tuple(Dimnames)
        # the Exporter throws std::invalid_argument on a missing slot (singlet.h:116-117)
        if self.p.shape[0] != self.Dim[1] + 1:
            raise ValueError("Cannot construct SparseMatrix from this object: length(p) != ncol + 1")
        if self.x.shape[0] != self.i.shape[0] or self.x.shape[0] != int(self.p[-1]):
            raise ValueError("Cannot construct SparseMatrix from this object: inconsistent x / i / p")

    @property
    def nrow(self):
        return self.Dim[0]

    @property
    def ncol(self):
        return self.Dim[1]

    @property
    def nnz(self):
        return int(self.p[-1])

    @classmethod
    def from_scipy(cls, M, Dimnames=(None, None)):
        M = M.tocsc()
        M.sort_indices()
        return cls(M.data, M.indices, M.indptr, M.shape, Dimnames)

    @classmethod
    def from_dense(cls, D):
        D = np.asarray(D, dtype=np.float64)
        nrow, ncol = D.shape
        xs, is_, p = [], [], [0]
        for c in range(ncol):
            r = np.nonzero(D[:, c])[0]
            is_.append(r.astype(np.int32))
            xs.append(D[r, c])
            p.append(p[-1] + r.size)
        return cls(np.concatenate(xs) if xs else [], np.concatenate(is_) if is_ else [], p, (nrow, ncol))

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((self.x, self.i, self.p), shape=self.Dim)

    def col_slice(self, c0, c1):
        """Columns [c0, c1) as a new dgCMatrix (a shard of cells)."""
        s, e = int(self.p[c0]), int(self.p[c1])
        return dgCMatrix(self.x[s:e], self.i[s:e], self.p[c0:c1 + 1] - self.p[c0], (self.Dim[0], c1 - c0))


def as_dgCMatrix(A):
    """as(as(as(A, "dMatrix"), "generalMatrix"), "CsparseMatrix") of R/run_nmf.R:39."""
    if isinstance(A, dgCMatrix):
        return A
    if all(hasattr(A, s) for s in ("x", "i", "p")):
        dim = getattr(A, "Dim", None) or (A.nrow, A.ncol)
        return dgCMatrix(A.x, A.i, A.p, dim)
    if hasattr(A, "tocsc"):
        return dgCMatrix.from_scipy(A)
    raise TypeError("expected a dgCMatrix-like object or a scipy sparse matrix")
