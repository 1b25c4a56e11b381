"""The generated NNLS solves (gen_nnls_lane.py, gen_nnls_half.py) checked WITHOUT a GPU: a small interpreter executes the
generated instruction stream of one sweep -- the same strings the kernels are assembled from -- on a few lanes with exact IEEE
arithmetic (C99 fma from libm, rationals as the fall-back), a Python loop around it plays the part of the kernel body (the stop test, the gates, tol), and
the result is held against the oracle's nnls (src/singlet.cpp:229-250 restated) column by column: same sweep counts, same zero
pattern, solutions within 1e-9.  What this pins down on every CPU run: the register plans (no overlap, ring slots, x in the
accumulator file), which half owns which coordinate and what the gates do, the broadcasts across the halves, the hand-over of tol,
the order of the row-update FMAs against the reads that refill their operands.  (What it cannot see -- wait states, LDS timing --
is asserted on the assembly in tests/test_kernel_codegen.py and on the device in tests/test_gpu_ops.py.)"""
import importlib.util
import math
import os
import re
import struct
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "singlet_amd", "csrc")


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(CSRC, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    argv = list(__import__("sys").argv)
    __import__("sys").argv = [name]          # the generators read their instance list from argv at import
    try:
        spec.loader.exec_module(mod)
    finally:
        __import__("sys").argv = argv
    return mod


try:   # C99 fma: correctly rounded, in hardware or in libm
    import ctypes
    import ctypes.util
    _libm_fma = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6").fma
    _libm_fma.restype = ctypes.c_double
    _libm_fma.argtypes = [ctypes.c_double] * 3
    assert _libm_fma(2.0 ** 53 + 2, 2.0 ** -53, -1.0) == 2.0 ** -52     # (a rounded product would give 2^-52 too; a fused one must)
    assert _libm_fma(1.0 + 2.0 ** -52, 1.0 + 2.0 ** -52, -1.0) == 2.0 ** -51 + 2.0 ** -104
except Exception:   # pragma: no cover
    _libm_fma = None


def _fma(a, b, c):
    if _libm_fma is not None:
        return _libm_fma(a, b, c)
    if hasattr(math, "fma"):
        return math.fma(a, b, c)
    if not (math.isfinite(a) and math.isfinite(b) and math.isfinite(c)):
        return a * b + c
    r = Fraction(a) * Fraction(b) + Fraction(c)
    if r == 0:   # the sign of an exact zero sum: round-to-nearest gives +0 unless both addends are -0
        p = a * b
        return p + c if (p == 0 and c == 0) else 0.0
    return float(r)


def _fmin(a, b):   # v_min_f64: the number if one operand is NaN, -0 < +0
    if a != a:
        return b
    if b != b:
        return a
    if a == b:
        return a if math.copysign(1.0, a) < 0 else b
    return a if a < b else b


class Machine:
    """64 lanes, v0..v255 and a0..a255 as dwords; VALU instructions run on `lanes` only (every lane is independent but for the
    DPP row broadcast of the Gram operand and the swap across the halves), LDS reads on all of them."""

    def __init__(self, lanes, lds, ops):
        self.v = np.zeros((512, 64), dtype=np.uint32)
        self.s = {}
        self.vcc = 0
        self.lanes = list(lanes)
        self.lds = lds            # bytes
        self.ops = ops            # name -> float | int | per-lane array

    # ---- registers
    def rd64(self, r, lane):
        return struct.unpack("<d", struct.pack("<II", int(self.v[r, lane]), int(self.v[r + 1, lane])))[0]

    def wr64(self, r, lane, x):
        lo, hi = struct.unpack("<II", struct.pack("<d", x))
        self.v[r, lane], self.v[r + 1, lane] = lo, hi

    @staticmethod
    def vreg(tok):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return int(m.group(1))
        m = re.fullmatch(r"v(\d+)", tok)
        if m:
            return int(m.group(1))
        m = re.fullmatch(r"a\[(\d+):(\d+)\]", tok) or re.fullmatch(r"a(\d+)", tok)
        if m:
            return 256 + int(m.group(1))
        raise ValueError(tok)

    def src64(self, tok, lane):
        neg = tok.startswith("-")
        if neg:
            tok = tok[1:]
        ab = tok.startswith("|")
        if ab:
            tok = tok[1:-1]
        if tok.startswith("%["):
            x = float(self.ops[tok[2:-1]])
        elif tok[0] in "va":
            x = self.rd64(self.vreg(tok), lane)
        else:
            x = float(tok)
        if ab:
            x = abs(x)
        return -x if neg else x

    def src32(self, tok, lane):
        if tok.startswith("%["):
            x = self.ops[tok[2:-1]]
            return int(x[lane]) if hasattr(x, "__len__") else int(x)
        if tok[0] in "va":
            return int(self.v[self.vreg(tok), lane])
        return int(tok, 0)

    def mask(self, tok):
        return self.vcc if tok == "vcc" else self.s.get(int(re.match(r"s\[(\d+):", tok).group(1)), 0)

    def set_mask(self, tok, m):
        if tok == "vcc":
            self.vcc = m
        else:
            self.s[int(re.match(r"s\[(\d+):", tok).group(1))] = m

    # ---- one instruction
    def run(self, ins):
        op, _, rest = ins.partition(" ")
        if op in ("s_waitcnt", "s_nop"):
            return
        dpp = None
        if op == "v_fmac_f64_dpp":
            rest, tail = rest.split(" row_newbcast:")
            dpp = int(tail.split()[0])
        a = [t.strip() for t in rest.split(",")]
        if op.startswith("ds_read_b"):
            n = int(op[9:]) // 32
            m = re.fullmatch(r"(\S+) offset:(\d+)", a[1]) or re.fullmatch(r"(\S+)", a[1])
            off = int(m.group(2)) if m.lastindex == 2 else 0
            d = self.vreg(a[0])
            for lane in range(64):
                addr = self.src32(m.group(1), lane) + off
                self.v[d:d + n, lane] = np.frombuffer(self.lds, dtype=np.uint32, count=n, offset=addr)
            return
        if op == "v_permlane32_swap_b32":
            x, y = self.vreg(a[0]), self.vreg(a[1])
            up = self.v[x, 32:].copy()
            self.v[x, 32:] = self.v[y, :32]
            self.v[y, :32] = up
            return
        if op == "s_and_b64":
            self.set_mask(a[0], self.mask(a[1]) & self.mask(a[2]))
            return
        if op.startswith("v_cmp_"):
            rel = {"lt": lambda p, q: p < q, "neq": lambda p, q: p != q}[op.split("_")[2]]
            m = 0
            for lane in self.lanes:
                if rel(self.src64(a[1], lane), self.src64(a[2], lane)):
                    m |= 1 << lane
            self.set_mask(a[0], m)
            return
        for lane in self.lanes:
            if op == "v_mul_f64":
                self.wr64(self.vreg(a[0]), lane, self.src64(a[1], lane) * self.src64(a[2], lane))
            elif op == "v_add_f64":
                self.wr64(self.vreg(a[0]), lane, self.src64(a[1], lane) + self.src64(a[2], lane))
            elif op == "v_fma_f64":
                self.wr64(self.vreg(a[0]), lane, _fma(self.src64(a[1], lane), self.src64(a[2], lane), self.src64(a[3], lane)))
            elif op == "v_min_f64":
                self.wr64(self.vreg(a[0]), lane, _fmin(self.src64(a[1], lane), self.src64(a[2], lane)))
            elif op == "v_rcp_f64":   # (the hardware's is an approximation; the two Newton steps behind it end on the same double)
                self.wr64(self.vreg(a[0]), lane, 1.0 / self.src64(a[1], lane))
            elif op == "v_fmac_f64_dpp":
                d, g = self.vreg(a[0]), self.vreg(a[1])
                self.wr64(d, lane, _fma(self.rd64(g, (lane & ~15) + dpp), self.src64(a[2], lane), self.rd64(d, lane)))
            elif op == "v_cndmask_b32_e64":
                sel = (self.mask(a[3]) >> lane) & 1
                self.v[self.vreg(a[0]), lane] = self.src32(a[2] if sel else a[1], lane)
            elif op in ("v_mov_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"):
                self.v[self.vreg(a[0]), lane] = self.src32(a[1], lane)
            else:
                raise NotImplementedError(ins)


def _problem(k, ncols, seed, L1):
    rng = np.random.default_rng(seed)
    F = rng.random((3 * k + 5, k))
    G = F.T @ F + 1e-15 * np.eye(k)
    B = (rng.normal(size=(ncols, k)) * 3 + 1.0) * np.exp(rng.normal(size=(ncols, 1)))
    X0 = np.abs(rng.normal(size=(ncols, k))) * (rng.random((ncols, k)) < 0.5) * 1e-3
    return G, B, X0


def _padded(G, k, KP):
    Gp = np.zeros((KP, KP))
    Gp[:k, :k] = G
    d = np.ones(KP)
    d[:k] = 1.0 / np.diag(G)          # (correctly rounded reciprocals: pad_gram_kernel)
    return Gp, d


def _solve(m, sweep, k, cols, tol_reg, set_gates, x_of):
    """the kernel body around the sweep: go = it < 100 && tol / k > 1e-8, tol = 0 where a column iterates, the gates, it++"""
    it = {c: 0 for c in cols}
    for _ in range(101):
        go = {c: it[c] < 100 and (m.rd64(tol_reg, cols[c][0]) / float(k)) > 1e-8 for c in cols}
        if not any(go.values()):
            break
        for c, lanes in cols.items():
            if go[c]:
                for lane in lanes:
                    m.wr64(tol_reg, lane, 0.0)
        set_gates(go)
        for ins in sweep:
            m.run(ins)
        for c in cols:
            it[c] += 1 if go[c] else 0
    return {c: (x_of(c), it[c]) for c in cols}


def _check(res, ora, G, B, X0, L1, L2):
    for c, (x, it) in res.items():
        xo, _, ito = ora.nnls(G, B[c], X0[c], L1, L2)
        assert it == ito, (c, it, ito)
        assert np.array_equal(x == 0, xo == 0), c
        assert np.linalg.norm(x - xo) <= 1e-9 * max(np.linalg.norm(xo), 1e-300), c


@pytest.mark.parametrize("k,L1,L2", [(5, 0.0, 0.0), (6, 0.02, 0.01), (19, 0.01, 0.0), (50, 0.02, 0.0), (51, 0.0, 0.03), (64, 0.01, 0.01)])
def test_generated_lane_sweep_solves_like_the_oracle(ora, k, L1, L2):
    gen = _load("gen_nnls_lane")
    KP = (k + 1) // 2 * 2
    s = gen.Sweep(KP)
    sweep = list(s.build())
    NGP = s.NGP
    G, B, X0 = _problem(k, 3, 700 + k, L1)
    Gp, rd = _padded(G, k, KP)
    row = 16 * NGP
    lds = np.zeros(KP * row + 2 * KP)
    for i in range(KP):
        for l in range(16):
            for mm in range(s.NG):
                j = l + 16 * mm
                lds[i * row + l * NGP + mm] = Gp[i, j] if j < KP else 0.0
    lds[KP * row::2] = np.diag(Gp)
    lds[KP * row + 1::2] = rd
    lanes = [0, 1, 2] if k < 40 else [0, 1]
    gl = np.array([(lane & 15) * NGP * 8 for lane in range(64)], dtype=np.uint32)
    m = Machine(lanes, lds.tobytes(), {"gl": gl, "dl": KP * row * 8, "l1": L1, "l2": L2, "eps": 1e-15, "one_hi": 0x3ff00000})
    xr = (lambda j: 256 + 2 * j) if s.XA else s.x
    for lane in lanes:
        for j in range(k):
            m.wr64(s.b(j), lane, B[lane, j])
            m.wr64(xr(j), lane, X0[lane, j])
        m.wr64(s.TOL, lane, 1.0)

    def gates(go):
        for lane in lanes:
            m.wr64(s.GM, lane, 1.0 if go[lane] else 0.0)

    res = _solve(m, sweep, k, {c: [c] for c in lanes}, s.TOL, gates, lambda c: np.array([m.rd64(xr(j), c) for j in range(k)]))
    _check(res, ora, G, B, X0, L1, L2)


@pytest.mark.parametrize("k,L1,L2", [(65, 0.0, 0.0), (68, 0.02, 0.01), (97, 0.01, 0.0), (100, 0.02, 0.0), (103, 0.0, 0.02), (128, 0.01, 0.0)])
def test_generated_two_lane_solve_solves_like_the_oracle(ora, k, L1, L2):
    gen = _load("gen_nnls_half")
    KP = (k + 3) // 4 * 4
    s = gen.Sweep(KP)
    sweep = list(s.build())
    KH, NGHP = s.KH, s.NGHP
    G, B, X0 = _problem(k, 2, 900 + k, L1)
    Gp, rd = _padded(G, k, KP)
    row = 32 * NGHP
    lds = np.zeros(2 * KP + KP * row)
    lds[0:2 * KP:2] = np.diag(Gp)
    lds[1:2 * KP:2] = rd
    for i in range(KP):
        for h in range(2):
            for l in range(16):
                for mm in range(s.NGH):
                    jl = l + 16 * mm
                    lds[2 * KP + i * row + (h * 16 + l) * NGHP + mm] = Gp[i, h * KH + jl] if jl < KH else 0.0
    cols = {0: [0, 32], 1: [1, 33]} if k < 90 else {0: [0, 32]}      # (the large ranks: one column keeps the test in seconds)
    lanes = sorted(l for v in cols.values() for l in v)
    gl = np.array([2 * KP * 8 + ((lane >> 5) * 16 + (lane & 15)) * NGHP * 8 for lane in range(64)], dtype=np.uint32)
    m = Machine(lanes, lds.tobytes(), {"gl": gl, "dl": 0, "l1": L1, "l2": L2, "eps": 1e-15, "one_hi": 0x3ff00000})
    m.v[s.GL2, :] = gl + 64 * s.ROWB
    xr = (lambda j: 256 + 2 * j) if s.XA else s.x
    for c, (lo, hi) in cols.items():
        for j in range(k):
            lane, jj = (lo, j) if j < KH else (hi, j - KH)
            m.wr64(s.b(jj), lane, B[c, j])
            m.wr64(xr(jj), lane, X0[c, j])
        m.wr64(s.TOL, lo, 1.0)
        m.wr64(s.TOL, hi, 1.0)

    def gates(go):
        for c, (lo, hi) in cols.items():
            for lane in (lo, hi):
                m.wr64(s.GM[0], lane, 1.0 if go[c] and lane == lo else 0.0)
                m.wr64(s.GM[1], lane, 1.0 if go[c] and lane == hi else 0.0)

    def x_of(c):
        lo, hi = cols[c]
        return np.array([m.rd64(xr(j), lo) if j < KH else m.rd64(xr(j - KH), hi) for j in range(k)])

    res = _solve(m, sweep, k, cols, s.TOL, gates, x_of)
    for c, (lo, hi) in cols.items():   # both halves end a sweep on the same tol (the stop test must agree)
        assert m.rd64(s.TOL, lo) == m.rd64(s.TOL, hi)
    _check(res, ora, G, B, X0, L1, L2)
