#!/usr/bin/env python3
"""Stored entries per non-zero of the entry-stream layouts on the i.i.d. benchmark matrix (density p = 1 / 20), by Monte Carlo
over the per-(column, tile) counts Binomial(tile rows, p) -- the arithmetic of tiled_count_kernel (kernels_tiled.hip): a column
unit is padded to the longest of its columns, rounded up to groups of `grp` entry tuples, a chunk to whole 64-slot sets.

  pairs,  2 factors per lane (product, ranks 33 - 64): 32 units x 2 columns, tile rows (160 KiB - 512) / (8 KS)
  quads,  2 factors per lane (product, ranks <= 32):   32 units x 4 columns, 632-row tiles
  quads,  4 factors per lane (the lever DESIGN.md priced for ranks 33 - 64, round-4 verdict item 2): 16 units x 4 columns
          (4 x 2 accumulator registers per column unit: 16 units fill v[128:255]), LDS rows of 64 doubles -> 312-row tiles

Prints entries per non-zero and, from the rates measured on the part (profiles/README.md, round-4 ablations: 4.1 LDS cycles per
ds_read_b128 and CU, 4 cycles per VALU instruction and SIMD, 2.05 GHz), the LDS and VALU floors of a config-3 pass."""
import numpy as np

rng = np.random.default_rng(1)


def pad(TR, slots, units, grp=4, p=0.05, trials=20000):
    c = rng.binomial(TR, p, size=(trials, units, slots))
    g = (c.max(axis=2) + grp - 1) // grp
    gps = 64 // (grp * slots)                      # groups per 64-slot set: chunks are whole sets
    tot = g.sum(axis=1)
    tot = tot + ((gps - tot % gps) % gps)
    return float(tot.sum() * grp * slots) / float(c.sum())


NNZ, CLK = 1.5e9, 2.05e9
rows = [("pairs, 2 factors per lane, k = 50 (product)", pad(408, 2, 32), 2, 1, 2),
        ("pairs, 2 factors per lane, k = 64", pad(312, 2, 32), 2, 1, 2),
        ("quads, 4 factors per lane, k <= 64 (priced, not built)", pad(312, 4, 16), 4, 2, 4),
        ("quads, 4 factors per lane, groups of 2 tuples", pad(312, 4, 16, grp=2), 4, 2, 4),
        ("quads, 2 factors per lane, k <= 32 (product)", pad(632, 4, 32), 4, 1, 2)]
rows += [("ranks 65 - 128 today: TWO passes of pairs over k / 2 = 50 factors", 2 * pad(408, 2, 32), 2, 1, 2),
         ("ranks 65 - 128 in ONE pass: pairs, 4 factors per lane, 204-row tiles", pad(204, 2, 16), 2, 2, 4)]
print("%-58s %8s %10s %10s %12s" % ("layout", "entries", "LDS floor", "VALU floor", "instr / nnz"))
for name, e, slots, reads, fmas in rows:
    tuples = NNZ * e / slots
    lds_ms = tuples * reads * 4.1 / CLK / 256 * 1e3
    valu_ms = tuples * (1 + fmas) * 4.0 / CLK / 1024 * 1e3
    instr = e / slots * (1 + reads + fmas + 0.94)   # + the 0.94 scalar / wait / load instructions per tuple of the round-4 loop
    print("%-58s %8.4f %8.2f ms %8.2f ms %12.2f" % (name, e, lds_ms, valu_ms, instr))
