#!/usr/bin/env python3
"""Generates nnls_half_gen.inc: the hand-scheduled solve of the TWO-LANES-PER-COLUMN NNLS (kernels_nnls_half_asm.hip, round 5),
ranks 65 ... 128 against a shared Gram.  Companion of gen_nnls_lane.py (ranks up to 64: one lane per column).

Layout as in nnls_half.h: lane c (c < 32) and lane 32 + c of a wave share column c of the wave's 32 columns; the lower half-wave
holds coordinates 0 .. KH - 1 of b and x, the upper half KH .. 2 KH - 1, all in registers.  What differs from the compiled kernel:

  * ONE half computes a coordinate's step.  The compiled kernel sends b_i and x_i to both halves (four v_permlane32_swap and their
    copies) and runs the step twice; here both halves issue the step on their own registers -- the upper half's b[ii], x[ii] are
    coordinate KH + ii -- and the half that does NOT own coordinate i runs it under a zero gate (GML / GMU: 1.0 on the lanes of the
    owner whose column iterates, else 0.0), exactly as a stopped column does in the lane kernels: its diff is multiplied by 0, so
    nd = +-0, the clamp test is false, x and tol keep their bits.  Only nd_i crosses the halves: two copies + two swaps.
  * x_i <- x_i - nd and tol += |nd / (x_i + 1e-15)| are written fma(-gm, nd, x_i) and fma(|q|, gm, tol) on the broadcast nd: with
    gm = 1.0 the same single rounding as the reference's expressions, with gm = 0.0 the identity (all values finite).
  * tol is sequential over the coordinates (a clamp RESETS it to 1: src/singlet.cpp:240): the lower half accumulates coordinates
    0 .. KH - 1, hands its tol to the upper half at coordinate KH (one broadcast), the upper half's final value goes to both at the
    end of the sweep.
  * the schedule: coordinate i's KH row-update FMAs carry the head of coordinate i + 1's chain (b_{i+1}'s FMA is issued first) and
    the tail of coordinate i's, as in gen_nnls_lane.py; the compiled kernel runs chain and FMAs back to back between scheduling fences.

Same operations in the same order per column as nnls_half_kernel / the oracle for finite data: bit-identical results.  (A solve
that produces Inf / NaN -- not reachable from finite input with the 1e-15 ridge on the diagonal -- would spread it through the
zero gate; the compiled kernel stays the reference there and is what the library uses when L1 < 0.)

LDS (staged by the kernel): Gl[i][h][l][m] = G[i, h KH + l + 16 m] (l < 16, m < NGHP; a lane's piece of a row is contiguous: two
ds_read_b128), rows 0 .. 63 addressed from %[gl], rows from 64 on from a second base (the offset field has 16 bits);
Dl[i] = (G_ii, 1 / G_ii).  The row pieces rotate through a ring of three register quads (a row's second piece reuses the quad the
previous row's first piece leaves after its FMAs): 12 registers instead of two whole rows.

Registers v[V_T : 255]: T (chain temporaries, tol, the two gates, the second LDS base), D 4, the ring 12, B 2 KH, X 2 KH; above
KH = 50 x lives in a[0 : 2 KH - 1] (one wave per SIMD).  Operands: see kernels_nnls_half_asm.hip."""
import os
import sys

KPS = [int(a) for a in sys.argv[1:]] or [100]


def r2(b):
    return f"v[{b}:{b + 1}]"


def plan(KP):
    KH = KP // 2
    if KP % 4 or KH > 64 or KH < 34:
        raise SystemExit("gen_nnls_half.py: KP = %d is not an instance (multiples of 4, 68 ... 128)" % KP)
    NGH = (KH + 15) // 16
    NGHP = (NGH + 1) & ~1
    NP = NGHP // 2                      # b128 pieces per lane and row
    xa = 16 + 24 + 4 + 4 * (NP + 1) + 4 * KH > 256
    nt = 30 if xa else 24
    own = nt + 4 + 4 * (NP + 1) + (2 * KH if xa else 4 * KH)
    acc = 2 * KH if xa else 0
    V_T = 28 if xa else 16
    if V_T + own > 256:
        raise SystemExit("gen_nnls_half.py: KP = %d does not fit the register file" % KP)
    V_D = V_T + nt
    V_G = V_D + 4
    V_B = V_G + 4 * (NP + 1)
    V_X = V_B + 2 * KH
    TOP = V_X if xa else V_X + 2 * KH
    total = TOP + acc
    return dict(KP=KP, KH=KH, NGH=NGH, NGHP=NGHP, NP=NP, V_T=V_T, V_D=V_D, V_G=V_G, V_B=V_B, V_X=V_X, TOP=TOP, XA=xa, NT=nt,
                WAVES=max(1, min(2, 512 // ((total + 7) // 8 * 8))), ROWB=32 * NGHP * 8)


class Sweep:
    def __init__(self, KP):
        self.__dict__.update(plan(KP))
        T = self.V_T
        self.A, self.S = T, T + 2            # head temporaries (S: the Markstein remainder, then the copy the swap consumes)
        self.NDB = [T + 4, T + 6]            # nd_i as both halves see it, by coordinate parity
        self.DEN, self.R, self.E, self.Q = T + 8, T + 10, T + 12, T + 14
        self.TOL = T + 16
        self.GM = [T + 18, T + 20]           # gate of the lower / upper half's coordinates
        self.GL2 = T + 22                    # LDS base of rows 64 ...
        self.XH = [T + 24, T + 26]           # XA: x_i as the chain reads it (by coordinate parity), XN its new value
        self.XN = T + 28
        self.L = []

    def b(self, j):
        return self.V_B + 2 * j

    def x(self, j):
        return self.V_X + 2 * j

    def slot(self, i, p):
        return self.V_G + 4 * ((i * self.NP + p) % (self.NP + 1))

    def g(self, i, m):
        return self.slot(i, m >> 1) + 2 * (m & 1)

    def own(self, i):
        o = 1 if i >= self.KH else 0
        return o, i - o * self.KH

    def head(self, i):
        """chain of coordinate i up to the broadcast nd_i; reads b[ii], x[ii] of the owning half, D = (G_ii, 1 / G_ii)"""
        o, ii = self.own(i)
        A, S, NDB, GM = self.A, self.S, self.NDB[i & 1], self.GM[o]
        b, gii, rii = self.b(ii), self.V_D, self.V_D + 2
        x = self.XH[i & 1] if self.XA else self.x(ii)
        N = NDB if o else S                 # where v_min leaves nd: the swap's SECOND operand ends as the upper half's value
        C = S if o else NDB                 # ... its first as the lower half's
        ops = []
        if self.XA:
            ops += [f"v_accvgpr_read_b32 v{x}, a{2 * ii}", f"v_accvgpr_read_b32 v{x + 1}, a{2 * ii + 1}"]
        ops += [
            f"v_mul_f64 {r2(A)}, {r2(b)}, {r2(rii)}",
            f"v_fma_f64 {r2(S)}, -{r2(A)}, {r2(gii)}, {r2(b)}",
            f"v_fma_f64 {r2(A)}, {r2(S)}, {r2(rii)}, {r2(A)}",      # b_i / G_ii, correctly rounded (Markstein)
        ]
        if i + 1 < self.KP:
            ops.append(f"ds_read_b128 v[{self.V_D}:{self.V_D + 3}], %[dl] offset:{16 * (i + 1)}")
        ops += [
            f"v_add_f64 {r2(A)}, {r2(A)}, -%[l1]",
            f"v_fma_f64 {r2(A)}, %[l2], {r2(x)}, {r2(A)}",
            f"v_mul_f64 {r2(A)}, {r2(A)}, {r2(GM)}",                 # the other half, and a stopped column, take a zero step
            f"v_min_f64 {r2(N)}, -{r2(A)}, {r2(x)}",                 # nd = min(-diff, x_i)
            f"v_mov_b32 v{C}, v{N}",
            f"v_mov_b32 v{C + 1}, v{N + 1}",
            f"v_cmp_lt_f64_e64 s[{{c0}}:{{c0p}}], {r2(x)}, -{r2(A)}",  # -diff > x_i ...
            f"v_cmp_neq_f64_e32 vcc, 0, {r2(x)}",                    # ... and x_i != 0: the clamp from a positive value
            f"v_permlane32_swap_b32 v{S if o else NDB}, v{NDB if o else S}",
            f"v_permlane32_swap_b32 v{(S if o else NDB) + 1}, v{(NDB if o else S) + 1}",
            "s_and_b64 s[{c0}:{c0p}], vcc, s[{c0}:{c0p}]",
        ]
        return ops

    def tail(self, i):
        """the rest of coordinate i's chain on the broadcast nd: x_i <- x_i - gm nd, tol <- clamp ? 1 : tol + gm |nd / (x_i + 1e-15)|"""
        o, ii = self.own(i)
        ND, GM, DEN, R, E, Q = self.NDB[i & 1], self.GM[o], self.DEN, self.R, self.E, self.Q
        if self.XA:
            xo, x = self.XH[i & 1], self.XN
            first = [f"v_fma_f64 {r2(x)}, -{r2(GM)}, {r2(ND)}, {r2(xo)}", f"v_accvgpr_write_b32 a{2 * ii}, v{x}", f"v_accvgpr_write_b32 a{2 * ii + 1}, v{x + 1}"]
        else:
            x = self.x(ii)
            first = [f"v_fma_f64 {r2(x)}, -{r2(GM)}, {r2(ND)}, {r2(x)}"]
        return first + [
            f"v_add_f64 {r2(DEN)}, {r2(x)}, %[eps]",
            f"v_rcp_f64 {r2(R)}, {r2(DEN)}",
            f"v_fma_f64 {r2(E)}, -{r2(DEN)}, {r2(R)}, 1.0",
            f"v_fma_f64 {r2(R)}, {r2(R)}, {r2(E)}, {r2(R)}",
            f"v_fma_f64 {r2(E)}, -{r2(DEN)}, {r2(R)}, 1.0",
            f"v_fma_f64 {r2(R)}, {r2(R)}, {r2(E)}, {r2(R)}",
            f"v_mul_f64 {r2(Q)}, {r2(ND)}, {r2(R)}",
            f"v_fma_f64 {r2(E)}, -{r2(DEN)}, {r2(Q)}, {r2(ND)}",
            f"v_fma_f64 {r2(Q)}, {r2(E)}, {r2(R)}, {r2(Q)}",          # nd / (x_i + 1e-15), correctly rounded (sgl_div_normal)
            f"v_fma_f64 {r2(Q)}, |{r2(Q)}|, {r2(GM)}, {r2(self.TOL)}",
            "TOLSEL",
        ]

    def fma(self, i, j):
        return (f"v_fmac_f64_dpp {r2(self.b(j))}, {r2(self.g(i, j >> 4))}, {r2(self.NDB[i & 1])} "
                f"row_newbcast:{j & 15} row_mask:0xf bank_mask:0xf")

    def piece_read(self, i, p):
        base, off = ("%[gl]", self.ROWB * i) if i < 64 else (f"v{self.GL2}", self.ROWB * (i - 64))
        s = self.slot(i, p)
        return f"ds_read_b128 v[{s}:{s + 3}], {base} offset:{off + 16 * p}"

    # ---- emission with the two hazards hipcc would pad: trans result -> VALU read (1 wait state), VALU write -> permlane swap (2)
    def emit(self, s, cmp_slot=0):
        c0 = 40 + 2 * (cmp_slot & 1)
        if s == "TOLSEL":
            self.L.append(f"v_cndmask_b32_e64 v{self.TOL}, v{self.Q}, 0, s[{c0}:{c0 + 1}]")
            self.L.append(f"v_cndmask_b32_e64 v{self.TOL + 1}, v{self.Q + 1}, %[one_hi], s[{c0}:{c0 + 1}]")
            return
        s = s.replace("{c0}", str(c0)).replace("{c0p}", str(c0 + 1))
        if self.L and self.L[-1].startswith("v_rcp_f64") and r2(self.R) in s:
            self.L.append("s_nop 0")
        if s.startswith("v_permlane32_swap"):
            regs = {int(t.strip()[1:]) for t in s.split(None, 1)[1].split(",")}
            need = 0
            for back, prev in enumerate(reversed(self.L[-2:])):
                if prev.startswith("v_") and not prev.startswith("v_cmp"):
                    d = prev.split(None, 1)[1].split(",")[0].strip()       # destination: v17 or v[16:17]
                    lo, hi = (d[2:-1].split(":") if d.startswith("v[") else (d[1:], d[1:]))
                    if regs & set(range(int(lo), int(hi) + 1)):
                        need = max(need, 2 - back)
            if need:
                self.L.append(f"s_nop {need - 1}")
        self.L.append(s)

    def bcast(self, reg, want_upper):
        """reg <- the lower (upper) half's value of reg, in all lanes; S is free"""
        S = self.S
        self.emit(f"v_mov_b32 v{S}, v{reg}")
        self.emit(f"v_mov_b32 v{S + 1}, v{reg + 1}")
        self.emit(f"v_permlane32_swap_b32 v{S}, v{reg}")            # S = lower half's value, reg = upper half's
        self.emit(f"v_permlane32_swap_b32 v{S + 1}, v{reg + 1}")
        if not want_upper:
            self.emit(f"v_mov_b32 v{reg}, v{S}")
            self.emit(f"v_mov_b32 v{reg + 1}, v{S + 1}")

    def build(self):
        KP, KH, NP, L = self.KP, self.KH, self.NP, self.L
        hd_per_tl = int(os.environ.get("SGL_GEN_HALF_HEAD_PER_TAIL", "2"))   # head(i + 1) instructions per tail(i) instruction
        L.append(f"ds_read_b128 v[{self.V_D}:{self.V_D + 3}], %[dl]")
        for p in range(NP):
            L.append(self.piece_read(0, p))
        L.append("s_waitcnt lgkmcnt(0)")
        for op in self.head(0):
            self.emit(op, 0)
        for i in range(KP):
            # row i and the pair of i + 1 were requested at least half a block ago
            L.append("s_waitcnt lgkmcnt(0)")
            if i == KH:
                self.bcast(self.TOL, False)      # the lower half's tol goes on in the upper half
            if i + 1 < KP:
                L.append(self.piece_read(i + 1, 0))   # into the quad row i - 1's last piece left
            tl = [(op, i) for op in self.tail(i)]
            hd = [(op, i + 1) for op in self.head(i + 1)] if i + 1 < KP else []
            chain = []
            while tl or hd:
                for _ in range(hd_per_tl):
                    if hd:
                        chain.append(hd.pop(0))
                if tl:
                    chain.append(tl.pop(0))
            nxt = self.own(i + 1)[1] if i + 1 < KP else -1
            order = ([nxt] if nxt >= 0 else []) + [j for j in range(KH) if j != nxt]
            # pieces free up in order: after the last FMA on piece p, row i + 1's piece p + 1 takes its quad
            last_of_piece = {}
            for pos, j in enumerate(order):
                last_of_piece[j >> 5] = pos
            nch, done = len(chain), 0
            for pos, j in enumerate(order):
                L.append(self.fma(i, j))
                for p in range(NP - 1):
                    if last_of_piece.get(p) == pos and i + 1 < KP:
                        L.append(self.piece_read(i + 1, p + 1))
                want = (pos + 1) * nch // KH
                while done < want:
                    op, slot = chain[done]
                    self.emit(op, slot)
                    done += 1
            while done < nch:
                op, slot = chain[done]
                self.emit(op, slot)
                done += 1
        self.bcast(self.TOL, True)               # the sweep's tol, in both halves
        return L

    def text(self, L=None):
        return " \\\n".join(f'    "{ins}\\n\\t"' for ins in (L if L is not None else self.L))


def kernel_body(KP):
    s = Sweep(KP)
    s.build()
    sweep = s.L
    KH = s.KH
    L = []
    A = L.append
    go, sv, n0, t0, t1, c1 = "s[46:47]", "s[44:45]", "s48", "s49", "s50", "s[50:51]"
    R, E, Q, TOL = s.R, s.E, s.Q, s.TOL

    def go_mask():
        # go = valid && it < 100 && tol / k > 1e-8   (src/singlet.cpp:231; the quotient correctly rounded: sgl_div_normal)
        A(f"v_rcp_f64 {r2(R)}, %[kd]")
        A("s_nop 1")
        A(f"v_fma_f64 {r2(E)}, -%[kd], {r2(R)}, 1.0")
        A(f"v_fma_f64 {r2(R)}, {r2(R)}, {r2(E)}, {r2(R)}")
        A(f"v_fma_f64 {r2(E)}, -%[kd], {r2(R)}, 1.0")
        A(f"v_fma_f64 {r2(R)}, {r2(R)}, {r2(E)}, {r2(R)}")
        A(f"v_mul_f64 {r2(Q)}, {r2(TOL)}, {r2(R)}")
        A(f"v_fma_f64 {r2(E)}, -%[kd], {r2(Q)}, {r2(TOL)}")
        A(f"v_fma_f64 {r2(Q)}, {r2(E)}, {r2(R)}, {r2(Q)}")
        A("v_cmp_gt_u32_e32 vcc, 100, %[it]")
        A("s_nop 3")
        A(f"s_mov_b64 {c1}, vcc")
        A(f"v_cmp_lt_f64_e32 vcc, %[thr], {r2(Q)}")
        A("s_nop 3")
        A(f"s_and_b64 {go}, vcc, {c1}")
        A(f"s_and_b64 {go}, {go}, %[valid]")

    def per_lane(mask, fmt):
        """memory instruction fmt(j) for this lane's coordinates j < kh under `mask`: kh = KH in the lower half, k - KH in the upper
        (an instance serves k >= KP - 7: coordinates below KH - 7 exist in both halves)"""
        A(f"s_mov_b64 {sv}, exec")
        A(f"s_and_b64 exec, exec, {mask}")
        for j in range(KH):
            if j == max(0, KH - 7):
                A(f"s_mov_b64 {c1}, exec")
            if j >= KH - 7:
                A(f"s_mov_b64 exec, {c1}")
                A(f"v_cmp_lt_u32_e32 vcc, {j}, %[kh]")
                A("s_nop 3")
                A("s_and_b64 exec, exec, vcc")
            A(fmt(j))
        A(f"s_mov_b64 exec, {sv}")

    for r in range(s.V_B, s.TOP):
        A(f"v_mov_b32 v{r}, 0")
    if s.XA:
        for r in range(2 * KH):
            A(f"v_accvgpr_write_b32 a{r}, 0")
    A(f"v_add_u32 v{s.GL2}, {64 * s.ROWB}, %[gl]")
    xr = (lambda q: f"a[{2 * q}:{2 * q + 1}]") if s.XA else (lambda q: r2(s.x(q)))
    per_lane("%[valid]", lambda q: f"global_load_dwordx2 {r2(s.b(q))}, %[bp], off offset:{8 * q}")
    per_lane("%[valid]", lambda q: f"global_load_dwordx2 {xr(q)}, %[xp], off offset:{8 * q}")
    A("s_waitcnt vmcnt(0)")
    A(f"v_mov_b32 v{TOL}, %[lo]")
    A(f"v_mov_b32 v{TOL + 1}, %[hi]")
    go_mask()
    A(f"s_bcnt1_i32_b64 {n0}, {go}")            # lanes iterating at the start of this pass (two per column)
    A("2:")                                      # ---- sweep loop
    go_mask()
    A(f"s_cmp_eq_u64 {go}, 0")
    A("s_cbranch_scc1 3f")
    A("s_cmp_eq_u32 %[toend], 1")
    A("s_cbranch_scc1 4f")
    A(f"s_bcnt1_i32_b64 {t0}, {go}")            # re-pack the stragglers: leave the pass below 3 / 8 of the starters
    A(f"s_lshl_b32 {t0}, {t0}, 3")
    A(f"s_mul_i32 {t1}, {n0}, 3")
    A(f"s_cmp_lt_u32 {t0}, {t1}")
    A("s_cbranch_scc1 3f")
    A("4:")
    A("s_add_u32 %[ran], %[ran], 1")
    A(f"v_cndmask_b32_e64 v{TOL}, v{TOL}, 0, {go}")          # tol = 0 where the column iterates
    A(f"v_cndmask_b32_e64 v{TOL + 1}, v{TOL + 1}, 0, {go}")
    A("s_mov_b32 s50, s46")                                   # the gates: 1.0 on the iterating lanes of the lower / upper half
    A("s_mov_b32 s51, 0")
    A(f"v_mov_b32 v{s.GM[0]}, 0")
    A(f"v_cndmask_b32_e64 v{s.GM[0] + 1}, 0, %[one_hi], {c1}")
    A("s_mov_b32 s50, 0")
    A("s_mov_b32 s51, s47")
    A(f"v_mov_b32 v{s.GM[1]}, 0")
    A(f"v_cndmask_b32_e64 v{s.GM[1] + 1}, 0, %[one_hi], {c1}")
    L.extend(sweep)
    A(f"v_cndmask_b32_e64 v{s.A}, 0, 1, {go}")
    A(f"v_add_u32 %[it], %[it], v{s.A}")
    A("s_branch 2b")
    A("3:")                                      # ---- {go} = the lanes of the columns left unfinished (only when the pass re-packs)
    A(f"v_mov_b32 %[lo], v{TOL}")
    A(f"v_mov_b32 %[hi], v{TOL + 1}")
    per_lane("%[valid]", lambda q: f"global_store_dwordx2 %[xp], {xr(q)}, off offset:{8 * q}")
    per_lane(go, lambda q: f"global_store_dwordx2 %[bp], {r2(s.b(q))}, off offset:{8 * q}")
    A(f"s_mov_b64 %[um], {go}")
    A("s_waitcnt vmcnt(0)")
    return s, L


def main():
    out = ["// generated by gen_nnls_half.py -- do not edit", "#pragma once",
           "#define SGL_NNLS_HALF_ASM_INSTANCES(X_) " + " ".join(f"X_({KP})" for KP in KPS)]
    for KP in KPS:
        s, L = kernel_body(KP)
        p = plan(KP)
        out.append(f"// ---- KP = {KP}: T v{p['V_T']}, D v{p['V_D']}, ring v{p['V_G']}, B v{p['V_B']}, X v{p['V_X']}")
        out.append(f"#define NNLS_HALF_ASM_VT_{KP} {p['V_T']}")
        out.append(f"#define NNLS_HALF_ASM_NGHP_{KP} {p['NGHP']}")
        out.append(f"#define NNLS_HALF_ASM_NGH_{KP} {p['NGH']}")
        out.append(f"#define NNLS_HALF_ASM_WAVES_{KP} {p['WAVES']}")
        out.append(f"#define NNLS_HALF_ASM_THREADS_{KP} {512 if p['WAVES'] > 1 else 256}")
        out.append(f"#define NNLS_HALF_ASM_VCLOB_{KP} " + ", ".join([f'"v{r}"' for r in range(p['V_T'], p['TOP'])] +
                                                                     ([f'"a{r}"' for r in range(2 * p['KH'])] if p['XA'] else [])))
        out.append(f"#define NNLS_HALF_ASM_BODY_{KP} \\\n{s.text(L)}")
        out.append("")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
