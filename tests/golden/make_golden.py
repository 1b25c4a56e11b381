#!/usr/bin/env python3
"""Generates the committed golden vectors tests/golden/*.npz.

Run in the authoring container only (it needs nothing from /root/reference at run time:
the reference cannot be executed here -- no R / Rcpp / Eigen -- so the vectors come from
oracle/np_transcription.py, the independent numpy transcription of the reference lines,
and pin the C oracle and the HIP path against accidental drift).  Inputs are produced by the
hash generator (itself covered by the hand-derived KATs), so each file stores inputs AND
expected outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import np_transcription as npt  # noqa: E402


class M:
    pass


def synth(m, n, inv_density, seed=0x5EED):
    """SURVEY 8(d) generator in numpy (uint64 hash), genes x cells CSC."""
    levels = np.log1p(1.0 + np.arange(16, dtype=np.float64))
    cells = np.arange(n, dtype=np.uint64)[None, :]
    genes = np.arange(m, dtype=np.uint64)[:, None]
    nz = npt.draw_np(seed, cells, genes, inv_density)            # (m, n): rand(cell, gene)
    val = levels[((npt.rand_np(seed + 1, cells, genes) >> np.uint64(11)) % np.uint64(16)).astype(np.int64)]
    A = M()
    cols = [np.nonzero(nz[:, c])[0] for c in range(n)]
    A.i = np.concatenate(cols).astype(np.int32)
    A.p = np.concatenate([[0], np.cumsum([len(c) for c in cols])]).astype(np.int32)
    A.x = np.concatenate([val[cols[c], c] for c in range(n)])
    A.nrow, A.ncol = m, n
    return A


def transpose(A):
    D = np.zeros((A.nrow, A.ncol))
    for c in range(A.ncol):
        s = slice(A.p[c], A.p[c + 1])
        D[A.i[s], c] = A.x[s]
    T = M()
    rows = [np.nonzero(D[r, :])[0] for r in range(A.nrow)]
    T.i = np.concatenate(rows).astype(np.int32)
    T.p = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
    T.x = np.concatenate([D[r, rows[r]] for r in range(A.nrow)])
    T.nrow, T.ncol = A.ncol, A.nrow
    return T


def winit(k, m, seed=0x5EED):
    f = np.arange(k, dtype=np.uint64)[None, :]
    g = np.arange(m, dtype=np.uint64)[:, None]
    return ((npt.rand_np(seed + 2, f, g) >> np.uint64(11)).astype(np.float64) + 0.5) * 2.0 ** -53


def main():
    kats = [(123, 0, 0), (123, 1, 2), (123, 2, 1), (123, 999999, 29999), (2147483647, 5, 7), (1, 0, 1),
            (0, 0, 0), (2 ** 64 - 1, 2 ** 64 - 1, 2 ** 64 - 1), (42, 2 ** 40 + 3, 17)]
    np.savez(os.path.join(HERE, "rng_kat.npz"), args=np.array(kats, dtype=np.uint64),
             out=np.array([npt.rand_py(*a) for a in kats], dtype=np.uint64))

    A = synth(300, 400, 20)
    At = transpose(A)
    base = dict(Ax=A.x, Ai=A.i, Ap=A.p, Atx=At.x, Ati=At.i, Atp=At.p, dim=np.array([300, 400]))
    for tag, (k, L1, L2, it) in {"nmf_k8_l1_0": (8, 0.0, 0.0, 5), "nmf_k8_l1_01": (8, 0.01, 0.0, 5),
                                 "nmf_k8_l1_01_l2_01": (8, 0.01, 0.01, 5), "nmf_k30": (30, 0.01, 0.0, 3)}.items():
        w0 = winit(k, 300)
        r = npt.c_nmf(A, At, 0.0, it, L1, L1, L2, L2, w0)
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), w0=w0, L1=L1, L2=L2, maxit=it, w=r["w"], h=r["h"],
                            d=r["d"], tol=r["tol"], **base)
    w0 = winit(6, 300)
    r = npt.c_ard_nmf(A, At, 0.0, 5, 0.01, 0.0, w0, 77, 20, 1e-3, 2)
    np.savez_compressed(os.path.join(HERE, "ard_k6.npz"), w0=w0, L1=0.01, L2=0.0, maxit=5, seed=77, inv_density=20,
                        overfit_threshold=1e-3, trace_test_mse=2, w=r["w"], h=r["h"], d=r["d"], test_mse=r["test_mse"],
                        iter=r["iter"], tol=r["tol"], score_overfit=r["score_overfit"], **base)
    wp = np.random.default_rng(1).random((300, 5))
    r = npt.c_project_model(A, wp, 0.01, 0.0)
    np.savez_compressed(os.path.join(HERE, "project_k5.npz"), w=wp, L1=0.01, L2=0.0, h=r["h"], d=r["d"], **base)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
