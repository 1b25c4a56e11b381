#!/usr/bin/env python3
"""Generates acc_rows_gen.inc: the inner loop of the row-shared sparse accumulate (acc_rows_kernel.h).

One wave owns 64 columns; their k-vectors live in v[128:255] (column c in v[128+2c : 129+2c], lane = factor).
A chunk = (64-column block, row tile of F in LDS) is walked ROW-major: a *part* is one row of the chunk with
up to G = 2 of its non-zeros; the factor row F[:, r] is read ONCE per part (ds_read_b64, lane = factor) and
applied to both entries with an M0-indexed FP64 FMA:

    s_bfe_u32 / s_lshr_b32  m0, <packed column word>      ; M0[15:12] = DST-relative, M0[7:0] = 2 * column
    v_fmac_f64_dpp v[128:129], x, f  row_newbcast:slot     ; v[128 + M0 : 129 + M0] += x[slot] * f

Stream record ("packet", 192 B = 16 entry slots = 8 parts):
    x[16]     f64   slot values (0 for pads), slot = 2 * part + entry
    cw[8]     u32   column words: slots 2j (low half) and 2j + 1 (high half), each 0x8000 | 2 * column
    roff[8]   u32   byte offset of the part's row inside the LDS tile
A packet is fetched by two vector loads whose lanes read with (lane & 15): every 16-lane row of the wave gets
the same 16 values, which is what the DPP row broadcast of the FMA (x) and of the address add (roff) needs --
no cross-lane shuffling on the VALU.  DP packets are in flight per wave (ring positions i = 0..DP-1, counted
vmcnt).  The LDS read of part p of the NEXT packet is issued right after the second FMA of part p of this one
into the same register pair, so one buffer of 8 factor rows suffices and every read has 15 slots to land
(counted lgkmcnt: LDS returns in order, nothing else uses the counter inside the loop).

Register plan (asm-owned; the compiler is capped at v0..v63):
    v64 .. v64+2DP-1 ring: x;  the next DP registers: cw/roff;  then 8 LDS addresses;  v112..v127 8 factor rows
    s[88:95] the 8 column words of the running packet
"""
import sys

DP = 13
XP = [64 + 2 * i for i in range(DP)]          # register pairs must be even-aligned
MT = [64 + 2 * DP + i for i in range(DP)]
AD = [64 + 3 * DP + p for p in range(8)]
FB = [112 + 2 * p for p in range(8)]
assert AD[-1] < FB[0] and FB[-1] + 1 < 128
ACC = 128
S_PTR, S_NB = 84, 86
S_T = [88 + j for j in range(8)]
REC = 192


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

    def load(self, i, ahead):
        self.A(f"global_load_dwordx2 {r2(XP[i])}, %[xoff], s[{S_PTR}:{S_PTR + 1}] offset:{ahead * REC}")
        self.A(f"global_load_dword v{MT[i]}, %[moff], s[{S_PTR}:{S_PTR + 1}] offset:{ahead * REC + 128}")

    def addrs(self, i):
        """LDS addresses of the 8 parts of the packet at ring position i; M0 must be off"""
        if self.lds:
            for p in range(8):
                self.A(f"v_add_u32_dpp v{AD[p]}, v{MT[i]}, %[lane8] row_newbcast:{8 + p} row_mask:0xf bank_mask:0xf")

    def readlanes(self, i):
        if self.rdl:
            for j in range(8):
                self.A(f"v_readlane_b32 s{S_T[j]}, v{MT[i]}, {j}")

    def slots(self, i, nxt_reads):
        A = self.A
        for s in range(16):
            p = s >> 1
            if not (s & 1) and self.lds:
                A(f"s_waitcnt lgkmcnt({7 if nxt_reads else 7 - p})")   # this packet's factor row p
            if self.idx:
                A(f"s_lshr_b32 m0, s{S_T[p]}, 16" if s & 1 else f"s_bfe_u32 m0, s{S_T[p]}, 0x100000")
            f = FB[p] if self.lds else FB[0]
            if self.fma:
                A(f"v_fmac_f64_dpp {r2(ACC)}, {r2(XP[i])}, {r2(f)} row_newbcast:{s} row_mask:0xf bank_mask:0xf")
            if (s & 1) and nxt_reads and self.lds:
                A(f"ds_read_b64 {r2(FB[p])}, v{AD[p]}")   # part p of the next packet; the register pair is free now

    def packet(self, i, L_body, L_exit):
        """precondition: the LDS reads of this packet are issued (in order, nothing behind them)"""
        A = self.A
        n = (i + 1) % DP
        L_last, L_done = self.label("last"), self.label("done")
        A(f"{L_body[i]}:")
        A(f"s_cmp_le_u32 s{S_NB}, 1")
        A(f"s_cbranch_scc1 {L_last}")
        A(f"s_waitcnt vmcnt({2 * (DP - 2)})")   # this packet and the next have landed
        self.m0_off()
        self.addrs(n)
        self.readlanes(i)
        self.slots(i, True)
        A(f"s_branch {L_done}")
        A(f"{L_last}:")
        A(f"s_waitcnt vmcnt({2 * (DP - 1)})")
        self.readlanes(i)
        self.slots(i, False)
        A(f"{L_done}:")
        self.load(i, DP)
        A(f"s_add_u32 s{S_PTR}, s{S_PTR}, {REC}")
        A(f"s_addc_u32 s{S_PTR + 1}, s{S_PTR + 1}, 0")
        A(f"s_sub_u32 s{S_NB}, s{S_NB}, 1")
        A(f"s_cmp_eq_u32 s{S_NB}, 0")
        A(f"s_cbranch_scc1 {L_exit[i]}")
        if i == DP - 1:
            A(f"s_branch {L_body[0]}")

    def chunk(self):
        """whole chunk: %[nb] packets starting at ring position %[phase]; %[ptr] = the chunk's first packet"""
        A = self.A
        L_body = [self.label(f"b{i}") for i in range(DP)]
        L_pro = [self.label(f"p{i}") for i in range(DP)]
        L_exit = [self.label(f"x{i}") for i in range(DP)]
        L_end = self.label("end")
        A(f"s_mov_b64 s[{S_PTR}:{S_PTR + 1}], %[ptr]")
        A(f"s_mov_b32 s{S_NB}, %[nb]")
        A(f"s_mov_b32 s{S_T[0]}, 0")
        A(f"s_set_gpr_idx_on s{S_T[0]}, gpr_idx(DST)")
        self.m0_off()
        if not self.rdl:
            for j in range(8):
                A(f"s_mov_b32 s{S_T[j]}, 0x80048002")
        A(f"s_waitcnt vmcnt({2 * (DP - 1)})")   # the chunk's first packet is the oldest of the ring
        for i in range(1, DP):
            A(f"s_cmp_eq_u32 %[phase], {i}")
            A(f"s_cbranch_scc1 {L_pro[i]}")
        for i in range(DP):
            A(f"{L_pro[i]}:")
            self.addrs(i)
            if self.lds:
                for p in range(8):
                    A(f"ds_read_b64 {r2(FB[p])}, v{AD[p]}")
            A(f"s_branch {L_body[i]}")
        for i in range(DP):
            self.packet(i, L_body, L_exit)
        for i in range(DP):
            A(f"{L_exit[i]}:")
            A(f"s_mov_b32 %[phase], {(i + 1) % DP}")
            if i < DP - 1:
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
                     ("ACC_ROWS_CHUNK_ASM_NORDL", {"rdl": False}),
                     ("ACC_ROWS_CHUNK_ASM_NOLDSFMA", {"lds": False, "fma": False}),
                     ("ACC_ROWS_CHUNK_ASM_BARE", {"lds": False, "fma": False, "idx": False, "rdl": False})):
        g = Gen(**kw)
        g.chunk()
        out.append(f"#define {name} \\\n{g.text()}")
        out.append("")
    g = Gen()
    g.A(f"s_mov_b64 s[{S_PTR}:{S_PTR + 1}], %[ptr]")
    for i in range(DP):
        g.load(i, i)
    out.append(f"#define ACC_ROWS_RING_FILL_ASM \\\n{g.text()}")
    out.append("")
    g = Gen()
    for c in range(128):
        g.A(f"v_mov_b32 v{ACC + c}, 0")
    out.append(f"#define ACC_ROWS_ZERO_ASM \\\n{g.text()}")
    out.append("")
    # v64..v255 are outside the compiler's budget (acc_rows_kernel.h) and need no clobber
    clob = [f'"s{r}"' for r in range(S_PTR, S_T[-1] + 1)] + ['"memory"', '"scc"']
    out.append("#define ACC_ROWS_CLOBBERS " + ", ".join(clob))
    out.append(f"#define ACC_ROWS_REC_BYTES {REC}")
    out.append(f"#define ACC_ROWS_REC_SLOTS 16")
    out.append(f"#define ACC_ROWS_RING {DP}")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
