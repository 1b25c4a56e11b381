#!/usr/bin/env python3
"""Generates acc_rows_gen.inc: the inner loop of the row-shared sparse accumulate (kernels_rows.hip).

One wave owns 64 columns; their k-vectors live in v[128:255] (column c in v[128+2c : 129+2c], lane = factor).
A chunk = (64-column block, row tile of F in LDS) is walked ROW-major: a *part* is one row of the chunk with
up to G = 2 of its non-zeros; the factor row F[:, r] is read ONCE per part (ds_read_b64, lane = factor) and
applied to both entries with an M0-indexed FP64 FMA:

    s_bfe_u32 / s_lshr_b32  m0, <packed column word>      ; M0[15:12] = DST-relative, M0[7:0] = 2 * column
    v_fmac_f64_dpp v[128:129], x, f  row_newbcast:slot     ; v[128 + M0 : 129 + M0] += x[slot] * f

Stream record ("block", 768 B = 64 entry slots = 32 parts = 4 packets of 16 slots):
    x[64]      f64   slot values (0 for pads)
    cw[32]     u32   column words: slots 2i (low half) and 2i + 1 (high half), each 0x8000 | 2 * column
    roff[32]   u32   byte offset of the part's row inside the LDS tile
Four blocks are in flight per wave (ring slots d = 0..3, two loads each, counted vmcnt); the LDS reads of
packet g + 1 are issued before the FMAs of packet g (counted lgkmcnt: LDS returns in order, no SMEM inside).

Register plan (asm-owned, the compiler is limited to v0..v63):
    v64..71   x ring          v72..75  cw/roff ring     v76 cw of the running block   v77 temp
    v78, v79  replicated row offsets (parts 0-15, 16-31)
    v80..87   x of the 4 packets, replicated over the four 16-lane rows
    v88..95   LDS addresses   v96..127 two buffers of 8 factor rows
"""
import sys

D = 4
import os
G = int(os.environ.get('ROWS_G', '2'))   # entries per part (2: product candidate; 4: what-if for row parts of four)
PP = 16 // G                               # parts per packet
XB = [64 + 2 * d for d in range(D)]
CB = [72 + d for d in range(D)]
CC, T1, RA, RB = 76, 77, 78, 79
XQ = [80 + 2 * q for q in range(4)]
AD = [88 + p for p in range(8)]
FB = [[96 + 16 * b + 2 * p for p in range(8)] for b in range(2)]
ACC = 128
S_PTR, S_NB = 84, 86
S_T = [88 + j for j in range(8)]
REC = 768


def r2(b):
    return f"v[{b}:{b + 1}]"


class Gen:
    def __init__(self, lds=True, idx=True, fma=True, rdl=True):
        self.L = []
        self.lds, self.idx, self.fma, self.rdl = lds, idx, fma, rdl
        self.lab = 0

    def A(self, s):
        self.L.append(s)

    def label(self, stem):
        self.lab += 1
        return f".Lrows_{stem}_{self.lab}_%="

    def m0_off(self):
        self.A("s_mov_b32 m0, 0")

    def rep_roff(self, src):
        A = self.A
        A(f"v_mov_b32 v{T1}, v{src}")
        A(f"v_mov_b32 v{RA}, v{src}")
        A("s_nop 1")
        A(f"v_permlane32_swap_b32 v{T1}, v{RA}")
        A("s_nop 1")
        A(f"v_mov_b32 v{RB}, v{RA}")
        A("s_nop 1")
        A(f"v_permlane16_swap_b32 v{RA}, v{RB}")
        A("s_nop 1")

    def rep_x(self, d):
        A = self.A
        for w in (0, 1):
            A(f"v_mov_b32 v{XQ[0] + w}, v{XB[d] + w}")
            A(f"v_mov_b32 v{XQ[2] + w}, v{XB[d] + w}")
        A("s_nop 1")
        for w in (0, 1):
            A(f"v_permlane32_swap_b32 v{XQ[0] + w}, v{XQ[2] + w}")
        A("s_nop 1")
        for w in (0, 1):
            A(f"v_mov_b32 v{XQ[1] + w}, v{XQ[0] + w}")
            A(f"v_mov_b32 v{XQ[3] + w}, v{XQ[2] + w}")
        A("s_nop 1")
        for w in (0, 1):
            A(f"v_permlane16_swap_b32 v{XQ[0] + w}, v{XQ[1] + w}")
            A(f"v_permlane16_swap_b32 v{XQ[2] + w}, v{XQ[3] + w}")
        A("s_nop 1")

    def refill(self, d):
        A = self.A
        A(f"global_load_dwordx2 {r2(XB[d])}, %[xoff], s[{S_PTR}:{S_PTR + 1}]")
        A(f"global_load_dword v{CB[d]}, %[coff], s[{S_PTR}:{S_PTR + 1}] offset:512")
        A(f"s_add_u32 s{S_PTR}, s{S_PTR}, {REC}")
        A(f"s_addc_u32 s{S_PTR + 1}, s{S_PTR + 1}, 0")

    def addrs(self, q):
        """LDS addresses of the 8 parts of packet q (row offsets in RA / RB); M0 must be off"""
        if not self.lds:
            return
        R = RA if (q * PP) < 16 else RB
        for p in range(PP):
            self.A(f"v_add_u32_dpp v{AD[p]}, v{R}, %[lane8] row_newbcast:{(q * PP) % 16 + p} row_mask:0xf bank_mask:0xf")

    def reads(self, fb):
        if not self.lds:
            return
        for p in range(PP):
            self.A(f"ds_read_b64 {r2(FB[fb][p])}, v{AD[p]}")

    def slots(self, q, fb):
        A = self.A
        for s in range(16):
            j = s >> 1
            if self.idx:
                A(f"s_lshr_b32 m0, s{S_T[j]}, 16" if s & 1 else f"s_bfe_u32 m0, s{S_T[j]}, 0x100000")
            f = FB[fb][s // G] if self.lds else FB[0][0]
            if self.fma:
                A(f"v_fmac_f64_dpp {r2(ACC)}, {r2(XQ[q])}, {r2(f)} row_newbcast:{s} row_mask:0xf bank_mask:0xf")

    def readlanes(self, src, q):
        if self.rdl:
            for j in range(8):
                self.A(f"v_readlane_b32 s{S_T[j]}, v{src}, {8 * q + j}")

    def block(self, d, L_body, L_exit):
        """precondition: this block's row offsets replicated in RA / RB, the LDS reads of its packet 0 issued into
        buffer 0, its loads (ring slot d) landed"""
        A = self.A
        dn = (d + 1) % D
        A(f"{L_body[d]}:")
        self.m0_off()
        A(f"v_mov_b32 v{CC}, v{CB[d]}")
        self.rep_x(d)
        self.refill(d)
        for q in range(4):
            fb = q & 1
            if q < 3:
                if q > 0:
                    self.m0_off()
                self.addrs(q + 1)
                self.reads(fb ^ 1)
                if self.lds:
                    A(f"s_waitcnt lgkmcnt({PP})")   # packet q's factor rows; the reads just issued stay in flight
            else:
                L_np, L_join = self.label("np"), self.label("join")
                A(f"s_cmp_le_u32 s{S_NB}, 1")
                A(f"s_cbranch_scc1 {L_np}")
                A(f"s_waitcnt vmcnt({2 * (D - 1)})")   # the next block (ring slot dn) has landed
                self.m0_off()
                self.rep_roff(CB[dn])
                self.addrs(0)
                self.reads(0)
                if self.lds:
                    A(f"s_waitcnt lgkmcnt({PP})")
                A(f"s_branch {L_join}")
                A(f"{L_np}:")
                if self.lds:
                    A("s_waitcnt lgkmcnt(0)")
                A(f"{L_join}:")
            self.readlanes(CC, q)
            self.slots(q, fb)
        A(f"s_sub_u32 s{S_NB}, s{S_NB}, 1")
        A(f"s_cmp_eq_u32 s{S_NB}, 0")
        A(f"s_cbranch_scc1 {L_exit[d]}")
        if d == D - 1:
            A(f"s_branch {L_body[0]}")

    def chunk(self):
        """whole chunk: %[nb] blocks starting at ring slot %[phase]"""
        A = self.A
        L_body = [self.label(f"b{d}") for d in range(D)]
        L_pro = [self.label(f"p{d}") for d in range(D)]
        L_exit = [self.label(f"x{d}") for d in range(D)]
        L_end = self.label("end")
        A(f"s_mov_b64 s[{S_PTR}:{S_PTR + 1}], %[ptr]")
        A(f"s_mov_b32 s{S_NB}, %[nb]")
        A(f"s_mov_b32 s{S_T[0]}, 0")
        A(f"s_set_gpr_idx_on s{S_T[0]}, gpr_idx(DST)")
        self.m0_off()
        if not self.rdl:
            for j in range(8):
                A(f"s_mov_b32 s{S_T[j]}, 0x80048002")
        for d in range(1, D):
            A(f"s_cmp_eq_u32 %[phase], {d}")
            A(f"s_cbranch_scc1 {L_pro[d]}")
        for d in range(D):
            # chunk prologue at ring slot d: its loads are the oldest in flight
            A(f"{L_pro[d]}:")
            A(f"s_waitcnt vmcnt({2 * (D - 1)})")
            self.rep_roff(CB[d])
            self.addrs(0)
            self.reads(0)
            A(f"s_branch {L_body[d]}")
        for d in range(D):
            self.block(d, L_body, L_exit)
        for d in range(D):
            A(f"{L_exit[d]}:")
            A(f"s_mov_b32 %[phase], {(d + 1) % D}")
            if d < D - 1:
                A(f"s_branch {L_end}")
        A(f"{L_end}:")
        A("s_set_gpr_idx_off")
        A(f"s_mov_b64 %[ptr], s[{S_PTR}:{S_PTR + 1}]")

    def text(self):
        return " \\\n".join(f'    "{ins}\\n\\t"' for ins in self.L)


def main():
    out = ["// generated by gen_acc_rows.py -- do not edit", "#pragma once"]
    for name, kw in (("ACC_ROWS_CHUNK_ASM", {}), ("ACC_ROWS_CHUNK_ASM_NOLDS", {"lds": False}),
                     ("ACC_ROWS_CHUNK_ASM_NOIDX", {"idx": False}), ("ACC_ROWS_CHUNK_ASM_NOFMA", {"fma": False}),
                     ("ACC_ROWS_CHUNK_ASM_NORDL", {"rdl": False}), ("ACC_ROWS_CHUNK_ASM_NOLDSFMA", {"lds": False, "fma": False}),
                     ("ACC_ROWS_CHUNK_ASM_BARE", {"lds": False, "fma": False, "idx": False, "rdl": False})):
        g = Gen(**kw)
        g.chunk()
        out.append(f"#define {name} \\\n{g.text()}")
        out.append("")
    # initial ring fill: D blocks
    g = Gen()
    g.A(f"s_mov_b64 s[{S_PTR}:{S_PTR + 1}], %[ptr]")
    for d in range(D):
        g.refill(d)
    g.A(f"s_mov_b64 %[ptr], s[{S_PTR}:{S_PTR + 1}]")
    out.append(f"#define ACC_ROWS_RING_FILL_ASM \\\n{g.text()}")
    out.append("")
    g = Gen()
    for c in range(128):
        g.A(f"v_mov_b32 v{ACC + c}, 0")
    out.append(f"#define ACC_ROWS_ZERO_ASM \\\n{g.text()}")
    out.append("")
    # v64..v255 are outside the compiler's range (amdgpu_num_vgpr(64)) and need no clobber
    clob = [f'"s{r}"' for r in range(S_PTR, S_T[-1] + 1)] + ['"memory"', '"scc"']
    out.append("#define ACC_ROWS_CLOBBERS " + ", ".join(clob))
    out.append(f"#define ACC_ROWS_REC_BYTES {REC}")
    out.append(f"#define ACC_ROWS_RING {D}")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
