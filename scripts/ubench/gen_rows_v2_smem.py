#!/usr/bin/env python3
"""Generates acc_rows_gen.inc: the inner loop of the row-shared sparse accumulate (acc_rows_kernel.h).

One wave owns 64 columns; their k-vectors live in v[128:255] (column c in v[128+2c : 129+2c], lane = factor).
A chunk = (64-column block, row tile of F in LDS) is walked ROW-major: a *part* is one row of the chunk with
up to G = 2 of its non-zeros; the factor row F[:, r] is read ONCE per part (ds_read_b64, lane = factor) and
applied to both entries with an M0-indexed FP64 FMA:

    s_bfe_u32 / s_lshr_b32  m0, <packed column word>      ; M0[15:12] = DST-relative, M0[7:0] = 2 * column
    v_fmac_f64_dpp v[128:129], x, f  row_newbcast:slot     ; v[128 + M0 : 129 + M0] += x[slot] * f

Everything that is wave-uniform travels on the SCALAR side: the column words and the row offsets of a packet
(16 slots = 8 parts) are one 64-byte s_load_dwordx16 into one of four SGPR banks, three packets ahead; the
LDS address of a part is a plain v_add_u32 with the row offset as its scalar operand.  (v_readlane and the DPP
form of v_add_u32 measured at 7 - 10 and 5 cycles per wave-instruction on this part: they were the bulk of the
VALU time of the first version.)  Only x goes through vector loads (one coalesced 512-byte load per block,
four blocks in flight, replicated over the four 16-lane rows for the DPP broadcast).

Stream record ("block", 768 B = 64 entry slots = 32 parts = 4 packets):
    x[64]                 f64   slot values (0 for pads), slot = 16 * packet + 2 * part + entry
    4 x { cw[8], roff[8] } u32  per packet: column words (slots 2j | 2j + 1 << 16, each 0x8000 | 2 * column)
                                and byte offsets of the parts' rows inside the LDS tile
SMEM and LDS share lgkmcnt and SMEM returns out of order, so every packet ends with lgkmcnt(0); the LDS reads
of packet g + 1 are issued in the first eight M0 -> FMA gaps of packet g and have a packet's time to land.

Register plan (asm-owned; the compiler is capped at v0..v63):
    v64..71  x ring (4 blocks)     v72..79 x of the 4 packets, replicated     v80..87 LDS addresses
    v96..127 two buffers of 8 factor rows
    s[32:33] running block, s34 blocks left, s[36:99] four banks of 16 (cw[8], roff[8]), bank = packet index
    within its block -- SGPR state does not live across asm statements (the banks are re-primed per chunk).
"""
import sys

D = 4
XB = [64 + 2 * d for d in range(D)]
XQ = [72 + 2 * q for q in range(4)]
AD = [80 + p for p in range(8)]
FB = [[96 + 16 * b + 2 * p for p in range(8)] for b in range(2)]
ACC = 128
S_CUR, S_NB = 32, 34
BK = [36 + 16 * q for q in range(4)]
REC = 768


def r2(b):
    return f"v[{b}:{b + 1}]"


class Gen:
    def __init__(self, lds=True, idx=True, fma=True, smem=True):
        self.L = []
        self.lds, self.idx, self.fma, self.smem = lds, idx, fma, smem
        self.lab = 0

    def A(self, s):
        self.L.append(s)

    def label(self, stem):
        self.lab += 1
        return f".Lrows_{stem}_{self.lab}_%="

    def m0_off(self):
        self.A("s_mov_b32 m0, 0")

    def rep_x(self, d):
        """x of ring slot d = [R0 R1 R2 R3] (rows of 16 lanes = packets) -> XQ[q] = [Rq Rq Rq Rq]"""
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

    def load_x(self, d, blocks_ahead):
        self.A(f"global_load_dwordx2 {r2(XB[d])}, %[xoff], s[{S_CUR}:{S_CUR + 1}] offset:{blocks_ahead * REC}")

    def load_bank(self, packets_ahead, q):
        """bank of packet q + packets_ahead (q = packet index of the running block)"""
        if not self.smem:
            return
        g = q + packets_ahead
        self.A(f"s_load_dwordx16 s[{BK[g % 4]}:{BK[g % 4] + 15}], s[{S_CUR}:{S_CUR + 1}], {(g // 4) * REC + 512 + 64 * (g % 4)}")

    def addrs(self, q):
        """LDS addresses of the 8 parts of packet q (bank q % 4); M0 must be off"""
        if not self.lds:
            return
        for p in range(8):
            self.A(f"v_add_u32 v{AD[p]}, s{BK[q % 4] + 8 + p}, %[lane8]")

    def reads(self, fb):
        if not self.lds:
            return
        for p in range(8):
            self.A(f"ds_read_b64 {r2(FB[fb][p])}, v{AD[p]}")

    def slots(self, q, fb, nxt_reads=True):
        """the 16 entry slots of packet q (factor rows in buffer fb, column words in bank q); the first eight
        M0 -> FMA gaps carry the LDS reads of the next packet (into buffer fb ^ 1, addresses in v[AD])"""
        A = self.A
        for s in range(16):
            j = s >> 1
            if self.idx:
                if s & 1:
                    A(f"s_lshr_b32 m0, s{BK[q] + j}, 16")
                else:
                    A(f"s_bfe_u32 m0, s{BK[q] + j}, 0x100000")
            if s < 8 and nxt_reads and self.lds:
                A(f"ds_read_b64 {r2(FB[fb ^ 1][s])}, v{AD[s]}")
            f = FB[fb][j] if self.lds else FB[0][0]
            if self.fma:
                A(f"v_fmac_f64_dpp {r2(ACC)}, {r2(XQ[q])}, {r2(f)} row_newbcast:{s} row_mask:0xf bank_mask:0xf")

    def block(self, d, L_body, L_exit):
        """precondition: banks of packets 0..2 landed, the reads of packet 0 landed in buffer 0, the x loads of ring
        slots d.. are the only vector loads in flight"""
        A = self.A
        A(f"{L_body[d]}:")
        A(f"s_waitcnt vmcnt({D - 1})")   # x of this block (the oldest of the ring)
        self.m0_off()
        self.rep_x(d)
        self.load_x(d, D)
        for q in range(4):
            fb = q & 1
            if q > 0:
                self.m0_off()
            self.load_bank(3, q)
            if q < 3:
                self.addrs(q + 1)
                self.slots(q, fb)
            else:
                L_np, L_join = self.label("np"), self.label("join")
                A(f"s_cmp_le_u32 s{S_NB}, 1")
                A(f"s_cbranch_scc1 {L_np}")
                self.addrs(4)
                self.slots(q, fb)
                A(f"s_branch {L_join}")
                A(f"{L_np}:")
                self.slots(q, fb, nxt_reads=False)
                A(f"{L_join}:")
            A("s_waitcnt lgkmcnt(0)")
        A(f"s_add_u32 s{S_CUR}, s{S_CUR}, {REC}")
        A(f"s_addc_u32 s{S_CUR + 1}, s{S_CUR + 1}, 0")
        A(f"s_sub_u32 s{S_NB}, s{S_NB}, 1")
        A(f"s_cmp_eq_u32 s{S_NB}, 0")
        A(f"s_cbranch_scc1 {L_exit[d]}")
        if d == D - 1:
            A(f"s_branch {L_body[0]}")

    def chunk(self):
        """whole chunk: %[nb] blocks starting at ring slot %[phase]; %[ptr] = its first block"""
        A = self.A
        L_body = [self.label(f"b{d}") for d in range(D)]
        L_exit = [self.label(f"x{d}") for d in range(D)]
        L_end = self.label("end")
        A(f"s_mov_b64 s[{S_CUR}:{S_CUR + 1}], %[ptr]")
        A(f"s_mov_b32 s{S_NB}, %[nb]")
        for q in range(3):
            self.load_bank(0, q)
        A(f"s_mov_b32 s{BK[3]}, 0")
        A(f"s_set_gpr_idx_on s{BK[3]}, gpr_idx(DST)")
        self.m0_off()
        A("s_waitcnt lgkmcnt(0)")
        self.addrs(0)
        self.reads(0)
        A("s_waitcnt lgkmcnt(0)")
        for d in range(1, D):
            A(f"s_cmp_eq_u32 %[phase], {d}")
            A(f"s_cbranch_scc1 {L_body[d]}")
        for d in range(D):
            self.block(d, L_body, L_exit)
        for d in range(D):
            A(f"{L_exit[d]}:")
            A(f"s_mov_b32 %[phase], {(d + 1) % D}")
            if d < D - 1:
                A(f"s_branch {L_end}")
        A(f"{L_end}:")
        A("s_set_gpr_idx_off")
        A(f"s_mov_b64 %[ptr], s[{S_CUR}:{S_CUR + 1}]")

    def text(self):
        return " \\\n".join(f'    "{ins}\\n\\t"' for ins in self.L)


def main():
    out = ["// generated by gen_acc_rows.py -- do not edit", "#pragma once"]
    for name, kw in (("ACC_ROWS_CHUNK_ASM", {}), ("ACC_ROWS_CHUNK_ASM_NOLDS", {"lds": False}),
                     ("ACC_ROWS_CHUNK_ASM_NOIDX", {"idx": False}), ("ACC_ROWS_CHUNK_ASM_NOFMA", {"fma": False}),
                     ("ACC_ROWS_CHUNK_ASM_NOLDSFMA", {"lds": False, "fma": False}),
                     ("ACC_ROWS_CHUNK_ASM_BARE", {"lds": False, "fma": False, "idx": False})):
        g = Gen(**kw)
        g.chunk()
        out.append(f"#define {name} \\\n{g.text()}")
        out.append("")
    # ring fill at the wave's first block: x of D blocks
    g = Gen()
    g.A(f"s_mov_b64 s[{S_CUR}:{S_CUR + 1}], %[ptr]")
    for d in range(D):
        g.load_x(d, d)
    out.append(f"#define ACC_ROWS_RING_FILL_ASM \\\n{g.text()}")
    out.append("")
    g = Gen()
    for c in range(128):
        g.A(f"v_mov_b32 v{ACC + c}, 0")
    out.append(f"#define ACC_ROWS_ZERO_ASM \\\n{g.text()}")
    out.append("")
    # v64..v255 are outside the compiler's budget (acc_rows_kernel.h) and need no clobber
    clob = [f'"s{r}"' for r in range(S_CUR, BK[3] + 16)] + ['"memory"', '"scc"']
    out.append("#define ACC_ROWS_CLOBBERS " + ", ".join(clob))
    out.append(f"#define ACC_ROWS_REC_BYTES {REC}")
    out.append(f"#define ACC_ROWS_RING {D}")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
