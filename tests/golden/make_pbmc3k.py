#!/usr/bin/env python3
"""Decodes the reference's bundled data file data/pbmc3k.RData (the fixture of its only test,
tests/testthat/test-pbmc3k.R:1-7) into tests/golden/pbmc3k_counts.npz without R.

The file is a bzip2-compressed RDX3 / XDR serialisation of a list {i, p, Dim, Dimnames, x =
rle{lengths, values}, cell_type}.  Only the matrix slots are kept (int counts, 13714 genes x
2700 cells, 2282976 non-zeros); row indices are delta-coded per column so the fixture stays small.
Run in the authoring container (reads /root/reference); the output is committed DATA.
"""
import bz2
import os
import struct
import sys

import numpy as np

SRC = "/root/reference/data/pbmc3k.RData"
HERE = os.path.dirname(os.path.abspath(__file__))


class R:
    def __init__(self, b):
        self.b, self.o = b, 0

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.o)[0]
        self.o += 4
        return v

    def vec_i32(self, n):
        v = np.frombuffer(self.b, dtype=">i4", count=n, offset=self.o).astype(np.int32)
        self.o += 4 * n
        return v

    def vec_f64(self, n):
        v = np.frombuffer(self.b, dtype=">f8", count=n, offset=self.o).astype(np.float64)
        self.o += 8 * n
        return v

    def length(self):
        n = self.i32()
        if n == -1:
            hi, lo = self.i32(), self.i32()
            n = (hi << 32) + lo
        return n


def read_item(r, refs):
    flags = r.i32()
    t = flags & 0xFF
    has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
    if t == 254:   # NILVALUE
        return None
    if t == 255:   # REFSXP
        idx = flags >> 8
        if idx == 0:
            idx = r.i32()
        return refs[idx - 1]
    if t == 1:     # SYMSXP
        name = read_item(r, refs)
        refs.append(name)
        return name
    if t == 2:     # LISTSXP (pairlist)
        out = []
        while True:
            attr = read_item(r, refs) if has_attr else None
            tag = read_item(r, refs) if has_tag else None
            car = read_item(r, refs)
            out.append((tag, car))
            flags = r.i32()
            t2 = flags & 0xFF
            if t2 == 254:
                break
            assert t2 == 2, t2
            has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
        return out
    if t == 9:     # CHARSXP
        n = r.i32()
        if n == -1:
            return None
        s = r.b[r.o:r.o + n].decode("utf-8", "replace")
        r.o += n
        return s
    if t == 13 or t == 10:    # INTSXP / LGLSXP
        v = r.vec_i32(r.length())
    elif t == 14:  # REALSXP
        v = r.vec_f64(r.length())
    elif t == 16:  # STRSXP
        v = [read_item(r, refs) for _ in range(r.length())]
    elif t == 19:  # VECSXP
        v = [read_item(r, refs) for _ in range(r.length())]
    else:
        raise ValueError("unsupported SEXP type %d at offset %d" % (t, r.o))
    attrs = read_item(r, refs) if has_attr else None
    if attrs and t == 19:
        names = [a[1] for a in attrs if a[0] == "names"]
        if names:
            return dict(zip(names[0], v))
    return v


def main():
    raw = bz2.decompress(open(SRC, "rb").read())
    assert raw[:5] == b"RDX3\n" and raw[5:7] == b"X\n", raw[:8]
    r = R(raw)
    r.o = 7
    r.i32(); r.i32(); r.i32()          # format version, writer R version, min reader version
    n = r.i32()                        # native encoding
    r.o += n
    top = read_item(r, [])             # pairlist of (symbol, value)
    obj = dict((k, v) for k, v in top)["pbmc3k"]
    i, p, dim = obj["i"], obj["p"], obj["Dim"]
    rle = obj["x"]
    x = np.repeat(np.asarray(rle["values"]), np.asarray(rle["lengths"]))
    assert tuple(dim) == (13714, 2700) and p[-1] == i.size == x.size == 2282976, (dim, p[-1], i.size, x.size)
    # delta-code rows inside each column
    di = i.copy()
    di[1:] -= i[:-1]
    di[p[:-1]] = i[p[:-1]]
    assert di.min() >= 0 and di.max() < 65536 and x.max() < 65536 and x.min() >= 1
    # rows of the mitochondrial genes (names "MT-..."): what the vignette's QC filter percent.mt needs
    # (docs/articles/Guided_Clustering_with_NMF.html: PercentageFeatureSet(pattern = "^MT-"))
    genes = obj["Dimnames"][0]
    mt_rows = np.array([q for q, g in enumerate(genes) if g.startswith("MT-")], dtype=np.int32)
    assert len(genes) == 13714 and 5 <= mt_rows.size <= 40, mt_rows.size
    np.savez_compressed(os.path.join(HERE, "pbmc3k_counts.npz"), di=di.astype(np.uint16), p=p.astype(np.int32),
                        x=x.astype(np.uint16), dim=np.asarray(dim, dtype=np.int32), mt_rows=mt_rows)
    print("pbmc3k: dim", tuple(dim), "nnz", i.size, "max count", int(x.max()), "MT genes", mt_rows.size)


if __name__ == "__main__":
    main()
