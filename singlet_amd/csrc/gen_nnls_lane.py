#!/usr/bin/env python3
"""Generates nnls_lane_gen.inc: the hand-scheduled sweep of the lane-per-column NNLS (kernels_nnls_asm.hip, round 5).

Why.  hipcc compiles a coordinate of nnls_lane_kernel (nnls_lane.h) as a strictly serial chain -- b_i / G_ii by the
Markstein form, the penalties, nd = min(-diff, x_i), the tol division: 22 dependent FP64 instructions -- FOLLOWED by the KP
independent row-update FMAs.  During the chain a wave uses a fraction of its VALU issue slots (every instruction waits for the
one before), and with two waves per SIMD (b and x take 4 KP registers) the FP64 pipe stands at 61 % busy.  The software-
pipelined form as C++ lost (round 3: 24 more live registers, scheduling fences).  Here the sweep is generated with its own
register plan, like the tiled accumulate's chunk loop:

  * coordinate i's row update  b_j += G_ji * nd_i  (KP DPP FMAs) carries, one chain instruction after every two FMAs,
    the TAIL of coordinate i's chain (x_i, the tol term: needs nd_i only) and the HEAD of coordinate i + 1's (up to nd_{i+1}:
    needs b_{i+1}, whose FMA is issued FIRST).  Dependent chain instructions are ~5 instructions apart: no issue slot waits;
  * the same operations in the same order per column as nnls_lane_kernel (sgl_nnls_nd / sgl_nnls_apply, sgl_div_normal):
    bit-identical results, sweep for sweep;
  * a padded coordinate (k = KP - 1) runs like the others: with b = x = 0, a zero Gram row and L1 >= 0 it is an exact no-op
    (nd = min(L1, 0) = 0, tol + |0 / 1e-15|), so there is no `i < k` test anywhere (the launcher keeps L1 < 0 off this path).

LDS (staged by the kernel): Gl row i = 16 NG doubles with lane l's entries l, l + 16, ... CONTIGUOUS (Gl[i][l][m] =
G[i, l + 16 m]): two ds_read_b128 at immediate offsets per row; Dl[i] = (G_ii, 1 / G_ii): one uniform ds_read_b128.

Registers v[V_T : 255], all clobbers of the ONE asm statement that holds a column's whole solve (load, sweep loop, store):
    T  20: 16 chain temporaries, TOL (the running tol), GM (1.0 / 0.0: the stopped-column gate of the sweep)
    D  4: (G_ii, 1 / G_ii) of the coming coordinate     G  2 x 2 NG: the row buffers     B  2 KP: b     X  2 KP: x
Operands: %[gl] / %[dl] (LDS byte addresses: this lane's row base, the diagonal table), %[one_hi] (a VGPR holding 0x3ff00000: a
second SGPR beside a select's mask would break the constant-bus limit), %[l1] %[l2] %[eps] %[kd] %[thr] (SGPR pairs: L1, L2,
1e-15, k, 1e-8), %[bp] %[xp] (the lane's column of B / X), %[valid] (lanes that own a column), %[it] %[lo] %[hi] (sweeps done,
tol: in / out), %[toend] %[klast] (run to the end / k == KP), %[ran] (sweeps executed, in / out), %[um] (out: lanes left
unfinished); s[40:51] and vcc are clobbered."""
import os
import sys

KPS = [int(a) for a in sys.argv[1:]] or [50]


def plan(KP):
    """Registers from the bottom: the compiler's v[0 : V_T - 1] (28, or what is left of 256 at the largest ranks: 16 at KP = 50),
    then T, D, G, B, X.  TOP = registers the kernel needs: small ranks leave room for more than two waves per SIMD.
    Above KP = 50 b and x no longer fit 256 registers: x then lives in the ACCUMULATOR half of the register file, a[0 : 2 KP - 1]
    (one wave per SIMD; a coordinate reads its x_i once and writes it once: two v_accvgpr moves each way), T grows by the
    copies of x_i the chain works on."""
    NG = (KP + 15) // 16
    NGP = (NG + 1) & ~1                 # doubles per lane and row in LDS: even, so that a lane's piece starts 16-byte aligned
    xa = 20 + 4 + 4 * NG + 4 * KP + 16 > 256
    nt = 26 if xa else 20
    own = nt + 4 + 4 * NG + (2 * KP if xa else 4 * KP)
    acc = 2 * KP if xa else 0
    if 16 + own > 256 or KP > 64:
        raise SystemExit("gen_nnls_lane.py: KP = %d does not fit the register file" % KP)
    # the compiler's share: 28 registers, or as few as 16 where that buys the second or third wave per SIMD (one wave issues an
    # FP64 instruction only every ~7.5 cycles, two reach 4.8 per SIMD: mix3 micro-benchmark; k = 30: nnls_h 0.515 -> 0.492 ms per
    # 200 000 cells with three; a fourth measured nothing at k = 20) or is all that is left (KP = 50)
    waves = lambda vt: max(1, min(8, 512 // ((vt + own + acc + 7) // 8 * 8)))
    V_T = max(range(16, min(28, 256 - own) + 1), key=lambda vt: (min(waves(vt), 3), vt))
    V_D = V_T + nt
    V_G = V_D + 4
    V_B = V_G + 4 * NG
    V_X = V_B + 2 * KP                  # (xa: unused -- x is a[0 : 2 KP - 1])
    TOP = V_X if xa else V_X + 2 * KP
    total = TOP + (2 * KP if xa else 0)
    return dict(KP=KP, NG=NG, NGP=NGP, V_T=V_T, V_D=V_D, V_G=V_G, V_B=V_B, V_X=V_X, TOP=TOP, XA=xa, NT=nt,
                WAVES=max(1, min(8, 512 // ((total + 7) // 8 * 8))))


def r2(b):
    return f"v[{b}:{b + 1}]"


class Sweep:
    def __init__(self, KP):
        p = plan(KP)
        self.__dict__.update(p)
        T = self.V_T
        self.A, self.E2 = T, T + 2          # head temporaries
        self.ND = [T + 4, T + 6]            # nd by coordinate parity
        self.DEN, self.R, self.E, self.Q = T + 8, T + 10, T + 12, T + 14
        self.TOL = T + 16                   # the column's running tol
        self.GM = T + 18                    # 1.0 / 0.0: the stopped-column gate of this sweep
        self.XH = [T + 20, T + 22]          # XA: x_i as the chain reads it (by coordinate parity), XN its new value
        self.XN = T + 24
        self.L = []

    def b(self, j):
        return self.V_B + 2 * j

    def x(self, j):
        return self.V_X + 2 * j

    def g(self, par, m):
        return self.V_G + 2 * self.NG * par + 2 * m

    def head(self, i):
        """chain of coordinate i up to nd_i: reads b_i, x_i, D = (G_ii, 1 / G_ii)"""
        A, E2, ND = self.A, self.E2, self.ND[i & 1]
        b, gii, rii = self.b(i), self.V_D, self.V_D + 2
        x = self.XH[i & 1] if self.XA else self.x(i)
        ops = []
        if self.XA:   # x_i out of the accumulator file
            ops += [f"v_accvgpr_read_b32 v{x}, a{2 * i}", f"v_accvgpr_read_b32 v{x + 1}, a{2 * i + 1}"]
        ops += [
            f"v_mul_f64 {r2(A)}, {r2(b)}, {r2(rii)}",
            f"v_fma_f64 {r2(E2)}, -{r2(A)}, {r2(gii)}, {r2(b)}",
            f"v_fma_f64 {r2(A)}, {r2(E2)}, {r2(rii)}, {r2(A)}",      # b_i / G_ii, correctly rounded (Markstein)
        ]
        if i + 1 < self.KP:   # D is free: the pair of the next coordinate
            ops.append(f"ds_read_b128 v[{self.V_D}:{self.V_D + 3}], %[dl] offset:{16 * (i + 1)}")
        ops += [
            f"v_add_f64 {r2(A)}, {r2(A)}, -%[l1]",
            f"v_fma_f64 {r2(A)}, %[l2], {r2(x)}, {r2(A)}",
            f"v_mul_f64 {r2(A)}, {r2(A)}, {r2(self.GM)}",            # a stopped column takes a zero step
            f"v_min_f64 {r2(ND)}, -{r2(A)}, {r2(x)}",                # nd = min(-diff, x_i)
            f"v_cmp_lt_f64_e64 s[{{c0}}:{{c0p}}], {r2(x)}, -{r2(A)}",  # -diff > x_i ...
            f"v_cmp_neq_f64_e32 vcc, 0, {r2(x)}",                    # ... and x_i != 0: the clamp from a positive value
            "s_and_b64 s[{c0}:{c0p}], vcc, s[{c0}:{c0p}]",
        ]
        return ops

    def tail(self, i):
        """the rest of coordinate i's chain: x_i <- x_i - nd, tol += |nd / (x_i + 1e-15)| or tol = 1"""
        ND, DEN, R, E, Q = self.ND[i & 1], self.DEN, self.R, self.E, self.Q
        if self.XA:
            xo, x = self.XH[i & 1], self.XN
            first = [f"v_add_f64 {r2(x)}, {r2(xo)}, -{r2(ND)}", f"v_accvgpr_write_b32 a{2 * i}, v{x}", f"v_accvgpr_write_b32 a{2 * i + 1}, v{x + 1}"]
        else:
            x = self.x(i)
            first = [f"v_add_f64 {r2(x)}, {r2(x)}, -{r2(ND)}"]
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
            f"v_add_f64 {r2(Q)}, {r2(self.TOL)}, |{r2(Q)}|",
            "TOLSEL",
        ]

    def fma(self, i, j):
        return (f"v_fmac_f64_dpp {r2(self.b(j))}, {r2(self.g(i & 1, j >> 4))}, {r2(self.ND[i & 1])} "
                f"row_newbcast:{j & 15} row_mask:0xf bank_mask:0xf")

    def row_reads(self, i):
        """row i of the Gram into buffer i & 1: this lane's 2 NG dwords-pairs are contiguous in LDS"""
        out, base = [], 16 * self.NGP * 8 * i
        m = 0
        while m < self.NG:
            if m + 1 < self.NG:
                out.append(f"ds_read_b128 v[{self.g(i & 1, m)}:{self.g(i & 1, m) + 3}], %[gl] offset:{base + 8 * m}")
                m += 2
            else:
                out.append(f"ds_read_b64 {r2(self.g(i & 1, m))}, %[gl] offset:{base + 8 * m}")
                m += 1
        return out

    def emit(self, s, cmp_slot):
        """cmp_slot: the coordinate whose compare result s[..] / TOLSEL this instruction belongs to (parity picks the SGPR pair)"""
        if s == "TOLSEL":
            c0 = 40 + 2 * (cmp_slot & 1)
            # tol = clamp-from-positive ? 1.0 : tol + |tadd|   (Q holds the sum)
            self.L.append(f"v_cndmask_b32_e64 v{self.TOL}, v{self.Q}, 0, s[{c0}:{c0 + 1}]")
            self.L.append(f"v_cndmask_b32_e64 v{self.TOL + 1}, v{self.Q + 1}, %[one_hi], s[{c0}:{c0 + 1}]")
            return
        c0 = 40 + 2 * (cmp_slot & 1)
        if self.L and self.L[-1].startswith("v_rcp_f64") and r2(self.R) in s:
            self.L.append("s_nop 0")   # transcendental result -> VALU read: one wait state (only where no FMA separates them)
        self.L.append(s.replace("{c0}", str(c0)).replace("{c0p}", str(c0 + 1)))

    def build(self):
        KP = self.KP
        L = self.L
        # prologue: the diagonal pair and the row of coordinate 0, its head un-interleaved
        L.append(f"ds_read_b128 v[{self.V_D}:{self.V_D + 3}], %[dl]")
        L.extend(self.row_reads(0))
        L.append("s_waitcnt lgkmcnt(0)")
        for op in self.head(0):
            self.emit(op, 0)
        for i in range(KP):
            # everything this block needs from LDS was requested a whole block ago (row i, the pair of i + 1)
            L.append("s_waitcnt lgkmcnt(0)")
            if i + 1 < KP:
                L.extend(self.row_reads(i + 1))
            chain = []
            tl = [(op, i) for op in self.tail(i)]
            hd = [(op, i + 1) for op in self.head(i + 1)] if i + 1 < KP else []
            # alternate head(i + 1) / tail(i): dependent instructions of one sub-chain end up ~5 instructions apart
            while tl or hd:
                if hd:
                    chain.append(hd.pop(0))
                if tl:
                    chain.append(tl.pop(0))
            order = ([i + 1] if i + 1 < KP else []) + [j for j in range(KP) if j != i + 1]   # b_{i+1} first: head(i + 1) waits for it
            nf = 0
            stride = int(os.environ.get("SGL_GEN_NNLS_STRIDE", "2"))   # FMAs per chain instruction (A/B builds; 2: 25 slots for 23)
            for j in order:
                L.append(self.fma(i, j))
                nf += 1
                if nf >= 2 and nf % stride == 0 and chain:
                    op, slot = chain.pop(0)
                    self.emit(op, slot)
            for op, slot in chain:   # (short ranks: more chain instructions than FMA pairs)
                self.emit(op, slot)
        return L

    def text(self, L=None):
        return " \\\n".join(f'    "{ins}\\n\\t"' for ins in (L if L is not None else self.L))


def kernel_body(KP):
    """Everything between the staging of the Gram and the epilogue bookkeeping as ONE asm statement: b and x then are plain
    clobbers of that statement (hipcc offers no way to keep it out of a register range ACROSS statements below 64 registers:
    amdgpu_num_vgpr is not honoured, waves_per_eu caps at 64).  Operands: see kernels_nnls_asm.hip."""
    s = Sweep(KP)
    s.build()
    sweep = s.L
    L = []
    A = L.append
    go, sv, n0, t0, t1, c1 = "s[46:47]", "s[44:45]", "s48", "s49", "s50", "s[50:51]"
    DEN, R, E, Q, TOL, GM = s.DEN, s.R, s.E, s.Q, s.TOL, s.GM

    def go_mask():
        # go = valid && it < 100 && tol / k > 1e-8   (src/singlet.cpp:231; the quotient correctly rounded: sgl_div_normal)
        A(f"v_rcp_f64 {r2(R)}, %[kd]")
        A("s_nop 1")     # a transcendental's result needs a wait state before a VALU instruction reads it (gfx940+)
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

    def masked(mask, body, last):
        A(f"s_mov_b64 {sv}, exec")
        A(f"s_and_b64 exec, exec, {mask}")
        L.extend(body)
        A("s_cmp_eq_u32 %[klast], 0")
        A("s_cbranch_scc1 1f")
        A(last)
        A("1:")
        A(f"s_mov_b64 exec, {sv}")

    for r in range(s.V_B, s.TOP):
        A(f"v_mov_b32 v{r}, 0")
    if s.XA:
        for r in range(2 * KP):
            A(f"v_accvgpr_write_b32 a{r}, 0")
    xr = (lambda q: f"a[{2 * q}:{2 * q + 1}]") if s.XA else (lambda q: r2(s.x(q)))
    j = KP - 1
    masked("%[valid]", [f"global_load_dwordx2 {r2(s.b(q))}, %[bp], off offset:{8 * q}" for q in range(KP - 1)],
           f"global_load_dwordx2 {r2(s.b(j))}, %[bp], off offset:{8 * j}")
    masked("%[valid]", [f"global_load_dwordx2 {xr(q)}, %[xp], off offset:{8 * q}" for q in range(KP - 1)],
           f"global_load_dwordx2 {xr(j)}, %[xp], off offset:{8 * j}")
    A("s_waitcnt vmcnt(0)")
    A(f"v_mov_b32 v{TOL}, %[lo]")
    A(f"v_mov_b32 v{TOL + 1}, %[hi]")
    go_mask()
    A(f"s_bcnt1_i32_b64 {n0}, {go}")            # columns iterating at the start of this pass
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
    A(f"v_mov_b32 v{GM}, 0")
    A(f"v_cndmask_b32_e64 v{GM + 1}, 0, %[one_hi], {go}")    # gm = 1.0 / 0.0
    L.extend(sweep)
    A(f"v_cndmask_b32_e64 v{s.A}, 0, 1, {go}")
    A(f"v_add_u32 %[it], %[it], v{s.A}")
    A("s_branch 2b")
    A("3:")                                      # ---- {go} = the columns left unfinished (only when the pass re-packs)
    A(f"v_mov_b32 %[lo], v{TOL}")
    A(f"v_mov_b32 %[hi], v{TOL + 1}")
    masked("%[valid]", [f"global_store_dwordx2 %[xp], {xr(q)}, off offset:{8 * q}" for q in range(KP - 1)],
           f"global_store_dwordx2 %[xp], {xr(j)}, off offset:{8 * j}")
    masked(go, [f"global_store_dwordx2 %[bp], {r2(s.b(q))}, off offset:{8 * q}" for q in range(KP - 1)],
           f"global_store_dwordx2 %[bp], {r2(s.b(j))}, off offset:{8 * j}")
    A(f"s_mov_b64 %[um], {go}")
    A("s_waitcnt vmcnt(0)")
    return s, L


def main():
    out = ["// generated by gen_nnls_lane.py -- do not edit", "#pragma once",
           "#define SGL_NNLS_ASM_INSTANCES(X_) " + " ".join(f"X_({KP})" for KP in KPS)]
    for KP in KPS:
        s, L = kernel_body(KP)
        p = plan(KP)
        out.append(f"// ---- KP = {KP}: T v{p['V_T']}, D v{p['V_D']}, G v{p['V_G']}, B v{p['V_B']}, X v{p['V_X']}")
        out.append(f"#define NNLS_ASM_VT_{KP} {p['V_T']}")
        out.append(f"#define NNLS_ASM_NGP_{KP} {p['NGP']}")
        out.append(f"#define NNLS_ASM_WAVES_{KP} {p['WAVES']}")
        out.append(f"#define NNLS_ASM_VCLOB_{KP} " + ", ".join([f'"v{r}"' for r in range(p['V_T'], p['TOP'])] +
                                                                ([f'"a{r}"' for r in range(2 * KP)] if p['XA'] else [])))
        out.append(f"#define NNLS_ASM_BODY_{KP} \\\n{s.text(L)}")
        out.append("")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
