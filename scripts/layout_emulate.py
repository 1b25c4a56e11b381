#!/usr/bin/env python3
"""Host emulation of the entry-stream layout arithmetic (kernels_tiled.hip: tiled_count_kernel) -- stored entries per
non-zero for a CSC matrix, with the columns in matrix order or in descending-count order, pairs (p, 32 + p) or
neighbours (2p, 2p + 1).  Usage: layout_emulate.py [pbmc3k | iid M N INV] k"""
import sys
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def entries(p, i, nrow, k, order="sorted", pairing="adjacent"):
    ncol = p.shape[0] - 1
    KS = (k + 1) & ~1
    TR = (160 * 1024 - 512) // (KS * 8) // 8 * 8
    TR = min(TR, 984)
    T = (nrow + TR - 1) // TR
    col_of = np.repeat(np.arange(ncol), np.diff(p))
    cnt = np.zeros((ncol, T), dtype=np.int64)
    np.add.at(cnt, (col_of, i // TR), 1)
    nnzc = np.diff(p)
    perm = np.argsort(-nnzc, kind="stable") if order == "sorted" else np.arange(ncol)
    nwb = (ncol + 63) // 64
    padded = np.zeros((nwb * 64, T), dtype=np.int64)
    padded[:ncol] = cnt[perm]
    blk = padded.reshape(nwb, 64, T)
    if pairing == "adjacent":
        a, b = blk[:, 0::2, :], blk[:, 1::2, :]
    else:
        a, b = blk[:, :32, :], blk[:, 32:, :]
    g = (np.maximum(a, b) + 3) // 4          # groups per (wb, pair, t)
    tot = g.sum(axis=1)                      # per (wb, t)
    tot += (8 - tot % 8) % 8
    return int(tot.sum()) * 8


def main():
    a = sys.argv[1:]
    if a[0] == "pbmc3k":
        z = np.load(os.path.join(ROOT, "tests", "golden", "pbmc3k_counts.npz"))
        p, dim = z["p"].astype(np.int64), z["dim"]
        i = z["di"].astype(np.int64)          # row indices are delta-coded per column in the fixture
        for c in range(int(dim[1])):
            i[p[c]:p[c + 1]] = np.cumsum(i[p[c]:p[c + 1]])
        nrow, ncol = int(dim[0]), int(dim[1])
        k = int(a[1])
    else:
        m, n, inv, k = int(a[1]), int(a[2]), int(a[3]), int(a[4])
        rng = np.random.default_rng(0)
        D = rng.random((m, n)) < 1.0 / inv
        i = np.nonzero(D.T)[1].astype(np.int64)
        p = np.concatenate([[0], np.cumsum(D.sum(axis=0))]).astype(np.int64)
        nrow, ncol = m, n
    nnz = int(p[-1])
    # transpose
    col_of = np.repeat(np.arange(ncol), np.diff(p))
    o = np.argsort(i, kind="stable")
    ti = col_of[o]
    tp = np.concatenate([[0], np.cumsum(np.bincount(i, minlength=nrow))]).astype(np.int64)
    for name, (pp, ii, nr) in {"H side (columns = cells)": (p, i, nrow), "W side (columns = genes)": (tp, ti, ncol)}.items():
        for order in ("matrix", "sorted"):
            for pairing in ("p,32+p", "adjacent"):
                e = entries(pp, ii, nr, k, order, pairing)
                print("%-26s k=%-3d order=%-7s pairs=%-8s entries/nnz = %.3f" % (name, k, order, pairing, e / nnz))


if __name__ == "__main__":
    main()
