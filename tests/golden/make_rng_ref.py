#!/usr/bin/env python3
"""Generates tests/golden/rng_ref.npz from the REFERENCE's own `rng` class.

oracle/make_ref.sh compiles the class as it lies in /root/reference/src/singlet.cpp (lines 6-114, cut out at
build time into the git-ignored oracle/_ref/) behind a three-function C shim; this script calls that build and
stores inputs + the reference's outputs.  Authoring container only (needs /root/reference); the committed
fixture is data: (state, i, j) -> rand triples and draw grids.  It moves the integer part of the parity claim
(hash, mask indices, synthetic generator) from "hand-derived" to "reference-derived".
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
u64p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)

DRAW_DENSITIES = (1, 2, 3, 7, 20, 64, 1000)
DRAW_CELL0 = (0, 999000)
DRAW_NCELLS, DRAW_NGENES = 24, 1500


def ref_lib():
    subprocess.check_call(["sh", os.path.join(ROOT, "oracle", "make_ref.sh")])
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "librng_ref.so"))
    L.ref_rand2.argtypes = [C.c_uint64, u64p, u64p, C.c_int64, u64p]
    L.ref_draw_grid.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.c_uint64, C.c_int64, u8p]
    return L


def inputs():
    """(state, i, j) triples: random 64-bit, random in the index range the path uses, and edge values."""
    g = np.random.default_rng(20261002)
    edge = np.array([0, 1, 2, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32, 2 ** 63 - 1, 2 ** 63, 2 ** 64 - 1, 999999, 29999], dtype=np.uint64)
    ei, ej = np.meshgrid(edge, edge, indexing="ij")
    states = np.array([0, 1, 123, 0x5EED, 0x5EEE, 0x5EEF, 2147483647, 2 ** 64 - 1], dtype=np.uint64)
    blocks = []
    for s in states:
        n = 12500
        i = np.concatenate([ei.ravel(), g.integers(0, 2 ** 64, n // 2, dtype=np.uint64, endpoint=False),
                            g.integers(0, 1000000, n // 2, dtype=np.uint64)])
        j = np.concatenate([ej.ravel(), g.integers(0, 2 ** 64, n // 2, dtype=np.uint64, endpoint=False),
                            g.integers(0, 30000, n // 2, dtype=np.uint64)])
        blocks.append((np.full(i.shape, s, dtype=np.uint64), i, j))
    return [np.concatenate([b[q] for b in blocks]) for q in range(3)]


def main():
    L = ref_lib()
    state, i, j = inputs()
    out2 = np.empty_like(i)
    for s in np.unique(state):
        sel = np.nonzero(state == s)[0]
        a, b = np.ascontiguousarray(i[sel]), np.ascontiguousarray(j[sel])
        o2 = np.empty_like(a)
        L.ref_rand2(int(s), a.ctypes.data_as(u64p), b.ctypes.data_as(u64p), a.size, o2.ctypes.data_as(u64p))
        out2[sel] = o2
    grids = np.empty((len(DRAW_DENSITIES), len(DRAW_CELL0), DRAW_NCELLS, DRAW_NGENES), dtype=np.uint8)
    for a, inv in enumerate(DRAW_DENSITIES):
        for b, c0 in enumerate(DRAW_CELL0):
            buf = np.empty((DRAW_NCELLS, DRAW_NGENES), dtype=np.uint8)
            L.ref_draw_grid(42, inv, c0, DRAW_NCELLS, 0, DRAW_NGENES, buf.ctypes.data_as(u8p))
            grids[a, b] = buf
    path = os.path.join(HERE, "rng_ref.npz")
    np.savez_compressed(path, state=state, i=i, j=j, rand2=out2, draw_state=np.uint64(42),
                        draw_inv_density=np.array(DRAW_DENSITIES, dtype=np.uint64), draw_cell0=np.array(DRAW_CELL0, dtype=np.uint64),
                        draw=np.packbits(grids, axis=-1), draw_shape=np.array(grids.shape, dtype=np.int64))
    print("wrote %s: %d rand triples, %d draw grids of %d x %d" % (path, i.size, grids.shape[0] * grids.shape[1], DRAW_NCELLS, DRAW_NGENES))


if __name__ == "__main__":
    sys.exit(main())
