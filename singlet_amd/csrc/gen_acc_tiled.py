#!/usr/bin/env python3
"""Generates acc_tiled_gen.inc: the hand-scheduled chunk loop of the LDS-tiled accumulate (kernels_tiled.hip).

Same entry stream, same arithmetic and the same order of operations as the compiler-scheduled loop it replaces
(results are bit-identical); what changes is the schedule:

  * the LDS reads run ONE OCTET (8 entry pairs) ahead without ever draining: the read of pair j of the next octet
    is issued right after the two FMAs of pair j of this one, into the register quad those FMAs just consumed
    (one buffer of 8 quads; counted lgkmcnt(7): LDS returns in order and nothing else uses the counter in here).
    The compiler's schedule finished every 64-entry set with lgkmcnt(0) and paid the loaded LDS latency once
    per set;
  * the accumulators of the 32 column pairs stay IN PLACE in v[128:255]: VGPR index mode is on for the whole
    loop and M0 (destination-relative, 0x8000 | 4 * pair) is written before each group of FMAs and cleared before
    the address adds -- a pair switch is a dozen scalar instructions on a byte queue instead of eight moves through
    s_set_gpr_idx_on / off plus a v_readlane;
  * the next set is prepared (three v_permlane16_swap, stream refill) inside the last octet of the running one.

Stream (kernels_tiled.hip): sets of 64 entries = 32 of the A half, 32 of the B half; row offsets u32, values f64;
cnt[pair] = groups of 4 entry pairs of that column pair in the chunk (32 bytes per chunk, handed over in SGPRs).

Register plan (asm-owned; the compiler is capped at v0..v63 by amdgpu_waves_per_eu(8, 8)):
    v64..67 row-offset ring, v68..75 value ring (4 sets in flight, counted vmcnt)
    v76..81 / v82..87 prepared set (ra, rb, xa, xb), two copies   v88..95 LDS addresses   v96..127 8 factor quads
    v128..255 accumulators (pair p: v[128 + 4p : 131 + 4p])
    s[84:85] / s[86:87] next set to load (row offsets / values), s88 sets left, s89 groups left of the running
    pair, s90 running pair, s91 its M0 word, s[92:99] the chunk's group counts
"""
import os
import sys

ER = [64 + i for i in range(4)]
EX = [68 + 2 * i for i in range(4)]
P = [{"ra": 76 + 6 * b, "rb": 77 + 6 * b, "xa": 78 + 6 * b, "xb": 80 + 6 * b} for b in range(2)]
AD = [88 + j for j in range(8)]
W = [96 + 4 * j for j in range(8)]
ACC = 128
S_RP, S_XP, S_NS, S_REM, S_S, S_ACC, S_T = 84, 86, 88, 89, 90, 91, 83
S_Q = 92   # s[92:99]: the chunk's 32 group counts (one byte per column pair), consumed as a shift queue


def r2(b):
    return f"v[{b}:{b + 1}]"


class Gen:
    def __init__(self):
        self.L = []
        self.cold = []
        self.cold2 = []
        self.lab = 0

    def A(self, s):
        self.L.append(s)

    def label(self, stem):
        self.lab += 1
        return f".Ltiled_{stem}_{self.lab}_%="

    def refill(self, i):
        """ring slot i <- the set four ahead of the one it held.  s[S_RP] / s[S_XP] point at the set that goes into
        slot 0 of the current lap; slots are refilled in the order 0, 1, 2, 3, so the pointers advance once per lap
        (after slot 3) and the slot is an immediate offset: 2 loads per set, 4 scalar instructions per FOUR sets."""
        A = self.A
        A(f"global_load_dword v{ER[i]}, %[voff4], s[{S_RP}:{S_RP + 1}] offset:{256 * i}")
        A(f"global_load_dwordx2 {r2(EX[i])}, %[voff8], s[{S_XP}:{S_XP + 1}] offset:{512 * i}")
        if i == 3:
            A(f"s_add_u32 s{S_RP}, s{S_RP}, 1024")
            A(f"s_addc_u32 s{S_RP + 1}, s{S_RP + 1}, 0")
            A(f"s_add_u32 s{S_XP}, s{S_XP}, 2048")
            A(f"s_addc_u32 s{S_XP + 1}, s{S_XP + 1}, 0")

    def prep(self, i, p):
        """ring slot i -> prepared set p: rows [A0 A1 B0 B1] -> ra = [A0 A0 B0 B0], rb = [A1 A1 B1 B1] (same for x),
        then the slot is refilled with the set four ahead.  M0 must be off."""
        A = self.A
        A("s_waitcnt vmcnt(6)")   # the two loads of this slot are the oldest of the eight in flight
        A(f"v_mov_b32 v{p['ra']}, v{ER[i]}")
        A(f"v_mov_b32 v{p['rb']}, v{ER[i]}")
        for w in (0, 1):
            A(f"v_mov_b32 v{p['xa'] + w}, v{EX[i] + w}")
            A(f"v_mov_b32 v{p['xb'] + w}, v{EX[i] + w}")
        A("s_nop 1")
        A(f"v_permlane16_swap_b32 v{p['ra']}, v{p['rb']}")
        for w in (0, 1):
            A(f"v_permlane16_swap_b32 v{p['xa'] + w}, v{p['xb'] + w}")
        self.refill(i)
        A("s_nop 1")   # VALU write -> DPP read of the prepared registers

    def addrs(self, p, o):
        """LDS addresses of the 8 entry pairs of octet o (0..3) of prepared set p"""
        r = p["ra"] if o < 2 else p["rb"]
        for j in range(8):
            self.A(f"v_add_u32_dpp v{AD[j]}, v{r}, %[lane16] row_newbcast:{8 * (o & 1) + j} row_mask:0xf bank_mask:0xf")

    def group_head(self, set_m0=True):
        """before a group of 4 entry pairs: count the running pair's groups down; when they are used up take the next
        pair that has any -- its count is the low byte of the queue s[S_Q:S_Q+7] (32 bytes, shifted down one byte per
        pair, the next 64-bit word moved in after every eighth) -- and point M0 at its accumulators.  All scalar: a
        v_readlane here cost ~200 cycles per switch (the SALU waits for the VALU to drain)."""
        A = self.A
        L_sw, L_next, L_norot, L_go = self.label("sw"), self.label("nx"), self.label("nr"), self.label("go")
        A(f"s_sub_u32 s{S_REM}, s{S_REM}, 1")
        A(f"s_cbranch_scc1 {L_sw}")          # rare: the common path falls through (a taken branch per group cost ~10 %)
        if set_m0:
            A(f"{L_go}:")
            A(f"s_mov_b32 m0, s{S_ACC}")
        else:   # M0 still holds the running pair's word from the octet's first group: only a switch has to write it
            A(f"{L_go}:")
        B = self.cold.append                  # out of line, behind the loop
        B(f"{L_sw}:")
        B(f"{L_next}:")
        B(f"s_and_b32 s{S_REM}, s{S_Q}, 0xff")
        B(f"s_lshr_b64 s[{S_Q}:{S_Q + 1}], s[{S_Q}:{S_Q + 1}], 8")
        B(f"s_add_u32 s{S_S}, s{S_S}, 1")
        B(f"s_and_b32 s{S_T}, s{S_S}, 7")
        B(f"s_cmp_eq_u32 s{S_T}, 7")
        B(f"s_cbranch_scc0 {L_norot}")
        for w in range(3):
            B(f"s_mov_b64 s[{S_Q + 2 * w}:{S_Q + 2 * w + 1}], s[{S_Q + 2 * w + 2}:{S_Q + 2 * w + 3}]")
        B(f"{L_norot}:")
        B(f"s_cmp_eq_u32 s{S_REM}, 0")
        B(f"s_cbranch_scc1 {L_next}")
        B(f"s_sub_u32 s{S_REM}, s{S_REM}, 1")
        B(f"s_lshl_b32 s{S_ACC}, s{S_S}, 2")
        B(f"s_or_b32 s{S_ACC}, s{S_ACC}, 0x8000")
        if not set_m0:
            B(f"s_mov_b32 m0, s{S_ACC}")
        B(f"s_branch {L_go}")

    def octet_fmas(self, p, o, reads):
        """the 16 FMAs of octet o of prepared set p; reads: issue the next octet's read of pair j behind pair j"""
        A = self.A
        x = p["xa"] if o < 2 else p["xb"]
        for g in range(2):
            self.group_head(set_m0=(g == 0))
            # ONE wait per group: with at most 4 reads outstanding behind them, the four quads of this group have
            # landed (every instruction, s_waitcnt included, takes an issue slot of the wave: ~5 cycles)
            A(f"s_waitcnt lgkmcnt({4 if reads else 4 - 4 * g})")
            for j in range(4 * g, 4 * g + 4):
                bc = f"row_newbcast:{8 * (o & 1) + j} row_mask:0xf bank_mask:0xf"
                A(f"v_fmac_f64_dpp {r2(ACC)}, {r2(x)}, {r2(W[j])} {bc}")
                A(f"v_fmac_f64_dpp {r2(ACC + 2)}, {r2(x)}, {r2(W[j] + 2)} {bc}")
                if reads:
                    A(f"ds_read_b128 v[{W[j]}:{W[j] + 3}], v{AD[j]}")

    def body(self, d, L_body, L_exit):
        """one set at ring slot d; precondition: prepared in P[d & 1], the reads of its octet 0 issued"""
        A = self.A
        dn = (d + 1) % 4
        cur, nxt = P[d & 1], P[(d + 1) & 1]
        A(f"{L_body[d]}:")
        for o in range(3):
            A("s_mov_b32 m0, 0")
            self.addrs(cur, o + 1)
            self.octet_fmas(cur, o, True)
        L_last, L_done = self.label("last"), self.label("done")
        A("s_mov_b32 m0, 0")
        A(f"s_cmp_le_u32 s{S_NS}, 1")
        A(f"s_cbranch_scc1 {L_last}")
        self.prep(dn, nxt)
        self.addrs(nxt, 0)
        self.octet_fmas(cur, 3, True)
        A(f"{L_done}:")
        # the chunk's last octet (nothing to fetch ahead) lives out of line
        hot, self.L = self.L, []
        A(f"{L_last}:")
        self.octet_fmas(cur, 3, False)
        A(f"s_branch {L_done}")
        self.cold2 += self.L
        self.L = hot
        A(f"s_sub_u32 s{S_NS}, s{S_NS}, 1")
        A(f"s_cmp_eq_u32 s{S_NS}, 0")
        A(f"s_cbranch_scc1 {L_exit[d]}")
        if d == 3:
            A(f"s_branch {L_body[0]}")

    def chunk(self):
        """%[ns] sets starting at ring slot %[phase]"""
        A = self.A
        L_body = [self.label(f"b{d}") for d in range(4)]
        L_pro = [self.label(f"p{d}") for d in range(4)]
        L_exit = [self.label(f"x{d}") for d in range(4)]
        L_end = self.label("end")
        A(f"s_mov_b64 s[{S_RP}:{S_RP + 1}], %[rp]")
        A(f"s_mov_b64 s[{S_XP}:{S_XP + 1}], %[xp]")
        A(f"s_mov_b32 s{S_NS}, %[ns]")
        for w in range(4):
            A(f"s_mov_b64 s[{S_Q + 2 * w}:{S_Q + 2 * w + 1}], %[q{w}]")
        A(f"s_mov_b32 s{S_REM}, 0")
        A(f"s_mov_b32 s{S_S}, -1")
        A(f"s_mov_b32 s{S_ACC}, 0")
        A(f"s_set_gpr_idx_on s{S_ACC}, 0")   # index mode on, no operand indexed while M0[15:12] = 0
        A("s_mov_b32 m0, 0")
        for d in range(1, 4):
            A(f"s_cmp_eq_u32 %[phase], {d}")
            A(f"s_cbranch_scc1 {L_pro[d]}")
        for d in range(4):
            A(f"{L_pro[d]}:")
            self.prep(d, P[d & 1])
            self.addrs(P[d & 1], 0)
            for j in range(8):
                A(f"ds_read_b128 v[{W[j]}:{W[j] + 3}], v{AD[j]}")
            A(f"s_branch {L_body[d]}")
        for d in range(4):
            self.body(d, L_body, L_exit)
        for d in range(4):
            A(f"{L_exit[d]}:")
            A(f"s_mov_b32 %[phase], {(d + 1) % 4}")
            if d < 3:
                A(f"s_branch {L_end}")
        A(f"s_branch {L_end}")
        self.L += self.cold2 + self.cold   # last octets of a chunk, pair switches
        A(f"{L_end}:")
        A("s_mov_b32 m0, 0")
        A("s_set_gpr_idx_off")
        A(f"s_mov_b64 %[rp], s[{S_RP}:{S_RP + 1}]")
        A(f"s_mov_b64 %[xp], s[{S_XP}:{S_XP + 1}]")

    def text(self):
        return " \\\n".join(f'    "{ins}\\n\\t"' for ins in self.L)


class GenQuad(Gen):
    """Ranks up to 32: FOUR columns per LDS instruction.

    A lane holds two factors, so 16 lanes -- one DPP row -- cover k <= 32 and the four 16-lane rows of a wave work on
    four different columns (a column QUAD) at once: per entry quad one v_add_u32_dpp, one ds_read_b128 and two
    v_fmac_f64_dpp serve FOUR non-zeros where the pair layout serves two.  Stream: a set of 64 slots = 16 entries of each
    of the quad's four columns, row h of the wave holding column slot h -- exactly the operand layout row_newbcast
    wants, so there is no set preparation at all (the pair layout copies every set and swaps lane rows: 9 VALU per 64
    entries): the address adds and the FMAs read the ring registers themselves.  The LDS tile rows are 256 B apart
    (32 doubles whatever the rank): the 16-lane groups ds_read_b128 is served in take 8 lanes of one DPP row and 8 of
    its neighbour, i.e. bytes 0-63 + 192-255 of one tile row and 64-191 of another -- conflict-free exactly when all
    rows start on the same bank.  A wave owns 32 quads = 128 columns (accumulators in v[128:255] as before, group
    counts in the same 32-byte queue).

    Ring: 8 sets in flight (a set is consumed in two octets -- half the time of a pair-layout set -- and its slot can only
    be refilled when its last operand has been read): v64..71 row offsets, v72..87 values.  s_waitcnt vmcnt(12) before
    the first use of slot d + 1: the 14 loads of slots d + 1 .. d + 7 are in flight, its two are the oldest."""

    NS = 8
    QER = [64 + i for i in range(8)]
    QEX = [72 + 2 * i for i in range(8)]

    def refill(self, i):
        A = self.A
        A(f"global_load_dword v{self.QER[i]}, %[voff4], s[{S_RP}:{S_RP + 1}] offset:{256 * i}")
        A(f"global_load_dwordx2 {r2(self.QEX[i])}, %[voff8], s[{S_XP}:{S_XP + 1}] offset:{512 * i}")
        if i == self.NS - 1:
            A(f"s_add_u32 s{S_RP}, s{S_RP}, {256 * self.NS}")
            A(f"s_addc_u32 s{S_RP + 1}, s{S_RP + 1}, 0")
            A(f"s_add_u32 s{S_XP}, s{S_XP}, {512 * self.NS}")
            A(f"s_addc_u32 s{S_XP + 1}, s{S_XP + 1}, 0")

    def addrs(self, slot, o):
        """LDS addresses of the 8 entry quads of octet o (0, 1) of the set in ring slot `slot`"""
        for j in range(8):
            self.A(f"v_add_u32_dpp v{AD[j]}, v{self.QER[slot]}, %[lane16] row_newbcast:{8 * o + j} row_mask:0xf bank_mask:0xf")

    def octet_fmas(self, slot, o, reads):
        A = self.A
        x = self.QEX[slot]
        for g in range(2):
            self.group_head(set_m0=(g == 0))
            A(f"s_waitcnt lgkmcnt({4 if reads else 4 - 4 * g})")
            for j in range(4 * g, 4 * g + 4):
                bc = f"row_newbcast:{8 * o + j} row_mask:0xf bank_mask:0xf"
                A(f"v_fmac_f64_dpp {r2(ACC)}, {r2(x)}, {r2(W[j])} {bc}")
                A(f"v_fmac_f64_dpp {r2(ACC + 2)}, {r2(x)}, {r2(W[j] + 2)} {bc}")
                if reads:
                    A(f"ds_read_b128 v[{W[j]}:{W[j] + 3}], v{AD[j]}")

    def body(self, d, L_body, L_exit):
        """one set at ring slot d; precondition: its loads have landed, the reads of its octet 0 are issued"""
        A = self.A
        dn = (d + 1) % self.NS
        A(f"{L_body[d]}:")
        A("s_mov_b32 m0, 0")
        self.addrs(d, 1)
        self.octet_fmas(d, 0, True)
        L_last, L_done = self.label("last"), self.label("done")
        A("s_mov_b32 m0, 0")
        A(f"s_cmp_le_u32 s{S_NS}, 1")
        A(f"s_cbranch_scc1 {L_last}")
        A(f"s_waitcnt vmcnt({2 * (self.NS - 2)})")   # slot d + 1 has landed (d itself is refilled below: 2 (NS - 1) in flight)
        self.addrs(dn, 0)
        self.octet_fmas(d, 1, True)
        A(f"{L_done}:")
        hot, self.L = self.L, []
        A(f"{L_last}:")
        self.octet_fmas(d, 1, False)
        A(f"s_branch {L_done}")
        self.cold2 += self.L
        self.L = hot
        # every operand of slot d has been read (VALU issue is in order): fetch the set NS ahead into it
        self.refill(d)
        A(f"s_sub_u32 s{S_NS}, s{S_NS}, 1")
        A(f"s_cmp_eq_u32 s{S_NS}, 0")
        A(f"s_cbranch_scc1 {L_exit[d]}")
        if d == self.NS - 1:
            A(f"s_branch {L_body[0]}")

    def chunk(self):
        A = self.A
        n = self.NS
        L_body = [self.label(f"b{d}") for d in range(n)]
        L_pro = [self.label(f"p{d}") for d in range(n)]
        L_exit = [self.label(f"x{d}") for d in range(n)]
        L_end = self.label("end")
        A(f"s_mov_b64 s[{S_RP}:{S_RP + 1}], %[rp]")
        A(f"s_mov_b64 s[{S_XP}:{S_XP + 1}], %[xp]")
        A(f"s_mov_b32 s{S_NS}, %[ns]")
        for w in range(4):
            A(f"s_mov_b64 s[{S_Q + 2 * w}:{S_Q + 2 * w + 1}], %[q{w}]")
        A(f"s_mov_b32 s{S_REM}, 0")
        A(f"s_mov_b32 s{S_S}, -1")
        A(f"s_mov_b32 s{S_ACC}, 0")
        A(f"s_set_gpr_idx_on s{S_ACC}, 0")
        A("s_mov_b32 m0, 0")
        for d in range(1, n):
            A(f"s_cmp_eq_u32 %[phase], {d}")
            A(f"s_cbranch_scc1 {L_pro[d]}")
        for d in range(n):
            A(f"{L_pro[d]}:")
            A(f"s_waitcnt vmcnt({2 * (n - 1)})")   # all NS slots in flight, slot d the oldest
            self.addrs(d, 0)
            for j in range(8):
                A(f"ds_read_b128 v[{W[j]}:{W[j] + 3}], v{AD[j]}")
            A(f"s_branch {L_body[d]}")
        for d in range(n):
            self.body(d, L_body, L_exit)
        for d in range(n):
            A(f"{L_exit[d]}:")
            A(f"s_mov_b32 %[phase], {(d + 1) % n}")
            A(f"s_branch {L_end}")
        self.L += self.cold2 + self.cold
        A(f"{L_end}:")
        A("s_mov_b32 m0, 0")
        A("s_set_gpr_idx_off")
        A(f"s_mov_b64 %[rp], s[{S_RP}:{S_RP + 1}]")
        A(f"s_mov_b64 %[xp], s[{S_XP}:{S_XP + 1}]")


class GenPairRing(GenQuad):
    """The pair layout (two columns per LDS instruction, ranks 33 - 64) on the quad loop's structure (round 4): the
    stream format is unchanged -- sets of 64 slots = 32 entries of the A half, 32 of the B half -- but a ring slot now
    takes HALF a set, loaded with the lane rows doubled: lanes 0-15 and 16-31 both read the A half's entries
    16 s .. 16 s + 15, lanes 32-47 and 48-63 the B half's (duplicate addresses inside one load cost nothing: same cache
    lines).  That is the [A A B B] operand layout the DPP row broadcasts want, so the six v_mov + three
    v_permlane16_swap + waits that turned [A0 A1 B0 B1] into it for every set are gone; the address adds and the FMAs
    read the ring registers directly.  %[voff4] / %[voff8] carry the doubled lane mapping."""

    def refill(self, i):
        A = self.A
        st, sub = i >> 1, i & 1
        A(f"global_load_dword v{self.QER[i]}, %[voff4], s[{S_RP}:{S_RP + 1}] offset:{256 * st + 64 * sub}")
        A(f"global_load_dwordx2 {r2(self.QEX[i])}, %[voff8], s[{S_XP}:{S_XP + 1}] offset:{512 * st + 128 * sub}")
        if i == self.NS - 1:
            A(f"s_add_u32 s{S_RP}, s{S_RP}, {128 * self.NS}")
            A(f"s_addc_u32 s{S_RP + 1}, s{S_RP + 1}, 0")
            A(f"s_add_u32 s{S_XP}, s{S_XP}, {256 * self.NS}")
            A(f"s_addc_u32 s{S_XP + 1}, s{S_XP + 1}, 0")


S_TP = 92            # s[92:93]: next 64-byte block of the schedule table to load (even-aligned pair)
TAB0, TAB1 = 52, 68  # s[52:67] the running lap's 32 schedule words (u16), s[68:83] the next lap's


class GenTab(GenQuad):
    """Round 4: the column-unit bookkeeping as a SCHEDULE TABLE instead of a countdown.

    The counters say the pass is bound by instruction issue (each SIMD's two waves are "active" 96 % of the time between
    them; profiles/r4_pmc_sq_*.csv), and 1.35 of the 5.7 instructions per entry pair are scalar: a countdown + branch in
    front of every group of four entry tuples, M0 written every other group, ~15 instructions per switch of the column
    unit on a byte queue of group counts.  Now the stream builder writes one u16 per GROUP -- the M0 word of the column
    unit the group belongs to (0x8000 | 4 * unit: destination-relative index) -- and a group head is ONE instruction,
        s_bfe_u32 m0, s[TAB0 + g / 2], <16 bits at 16 (g & 1)>
    with g = 4 d + 2 o + group static in the unrolled lap of 8 ring slots x 2 octets x 2 groups = 32 groups = 64 bytes
    of table = one s_load_dwordx16.  No countdown, no branch, no switch code.  The next lap's words are loaded a lap
    ahead into TAB1 and moved down at the wrap of the ring (s_waitcnt lgkmcnt(0) there: SMEM returns out of order, so
    only a full drain proves it has landed; LDS waits in between stay correct with the scalar load outstanding -- the
    counter then only over-counts).  A chunk entered at ring slot `phase` loads the lap that contains its first group
    (table address of the chunk - 8 * phase bytes) and the one behind it.

    pair = True: two columns per LDS instruction (half-set ring slots, doubled lane rows: GenPairRing's loads);
    pair = False: four (GenQuad's)."""

    def __init__(self, pair, pf=0):
        super().__init__()
        self.pair = pair
        self.pf = pf   # > 0: at every wrap of the ring, touch the stream `pf` laps beyond the lap being loaded (below)

    def refill(self, i):
        if self.pair:
            GenPairRing.refill(self, i)
        else:
            GenQuad.refill(self, i)

    def addrs(self, slot, o):
        if "noadd" in os.environ.get("SGL_GEN_ABLATE", ""):
            return
        GenQuad.addrs(self, slot, o)

    def group_head_tab(self, d, o, g):
        gi = 4 * d + 2 * o + g
        self.A(f"s_bfe_u32 m0, s{TAB0 + gi // 2}, {hex((16 * (gi & 1)) | (16 << 16))}")

    def octet(self, d, o, reads):
        A = self.A
        x = self.QEX[d]
        abl = os.environ.get("SGL_GEN_ABLATE", "")   # timing ablations only (results are wrong): nofma, noread, noadd
        # LDS waits: one per W entry tuples (SGL_GEN_WAIT, default 4 = one per group).  Before tuple j the wave has 8 reads in
        # flight (8 - j of this octet, j of the next); the next W tuples need the W oldest: lgkmcnt(8 - W) -- without reads
        # ahead (a chunk's last octet) 8 - j - W.
        W_ = int(os.environ.get("SGL_GEN_WAIT", "4"))
        for g in range(2):
            self.group_head_tab(d, o, g)
            for j in range(4 * g, 4 * g + 4):
                if j % W_ == 0:
                    A(f"s_waitcnt lgkmcnt({8 - W_ if reads else 8 - j - W_})")
                bc = f"row_newbcast:{8 * o + j} row_mask:0xf bank_mask:0xf"
                if "nofma" not in abl:
                    A(f"v_fmac_f64_dpp {r2(ACC)}, {r2(x)}, {r2(W[j])} {bc}")
                    A(f"v_fmac_f64_dpp {r2(ACC + 2)}, {r2(x)}, {r2(W[j] + 2)} {bc}")
                if reads and "noread" not in abl:
                    A(f"ds_read_b128 v[{W[j]}:{W[j] + 3}], v{AD[j]}")

    def wrap(self):
        """wrap of the ring = end of a lap of 32 groups: the next lap's schedule words move down, the one behind is fetched"""
        A = self.A
        A("s_waitcnt lgkmcnt(0)")
        for w in range(8):
            A(f"s_mov_b64 s[{TAB0 + 2 * w}:{TAB0 + 2 * w + 1}], s[{TAB1 + 2 * w}:{TAB1 + 2 * w + 1}]")
        A(f"s_load_dwordx16 s[{TAB1}:{TAB1 + 15}], s[{S_TP}:{S_TP + 1}], 0x0")
        A(f"s_add_u32 s{S_TP}, s{S_TP}, 64")
        A(f"s_addc_u32 s{S_TP + 1}, s{S_TP + 1}, 0")
        if self.pf:
            # Stream prefetch into L2 (round 5).  The ring keeps one lap (8 slots) of the stream in flight per wave: a slot's data
            # must make the whole trip from HBM inside one lap of the loop (~3 us), and every load that takes longer stalls the
            # wave at its counted vmcnt.  Once per lap two more loads touch every 128-byte line of the lap `pf` laps beyond the
            # one being loaded (lane stride = lap bytes / 64; %[pfl] = lane * stride + pf * lap bytes of the row-offset stream,
            # the value stream is twice that): the ring's own loads of that lap then hit L2.  Destination v63 is a dummy:
            # nothing reads it, and both loads are older than ring slot 0's refill, so the vmcnt wait of body(NS - 1) -- a lap
            # later -- has seen them land before v63 is written again.
            A("s_mov_b32 m0, 0")   # index mode is on and M0 still points the last group's destinations at its accumulators
            A("v_lshlrev_b32 v63, 1, %[pfl]")
            A(f"global_load_dword v63, v63, s[{S_XP}:{S_XP + 1}]")
            A(f"global_load_dword v63, %[pfl], s[{S_RP}:{S_RP + 1}]")

    def body(self, d, L_body, L_last):
        """a half-set that is NOT the chunk's last one: both octets fetch ahead; s[S_NS] counts the fetching half-sets still to
        come and its borrow sends the flow to the last half-set's own code (no test inside the body)"""
        A = self.A
        dn = (d + 1) % self.NS
        A(f"{L_body[d]}:")
        A("s_mov_b32 m0, 0")
        self.addrs(d, 1)
        self.octet(d, 0, True)
        A("s_mov_b32 m0, 0")
        # slot d + 1 has landed: of the 2 (NS - 1) ring loads in flight its two are the oldest; with the two prefetch loads of
        # the last wrap in flight as well (issued between the refills of slot NS - 1 and slot 0) two more may stay
        # outstanding, except at d = NS - 1, where they are older than the slot waited for
        A(f"s_waitcnt vmcnt({2 * (self.NS - 2) + (2 if (self.pf and d != self.NS - 1) else 0)})")
        self.addrs(dn, 0)
        self.octet(d, 1, True)
        self.refill(d)
        if d == self.NS - 1:
            self.wrap()
        A(f"s_sub_u32 s{S_NS}, s{S_NS}, 1")
        A(f"s_cbranch_scc1 {L_last[dn]}")
        if d == self.NS - 1:
            A(f"s_branch {L_body[0]}")

    def last(self, d, L_last, L_end):
        """the chunk's last half-set at ring slot d: nothing of the next chunk is fetched (another tile will be in LDS)"""
        A = self.A
        A(f"{L_last[d]}:")
        A("s_mov_b32 m0, 0")
        self.addrs(d, 1)
        self.octet(d, 0, True)
        self.octet(d, 1, False)
        self.refill(d)
        A(f"s_mov_b32 %[phase], {(d + 1) % self.NS}")
        A(f"s_branch {L_end}")

    def chunk(self):
        A = self.A
        n = self.NS
        L_body = [self.label(f"b{d}") for d in range(n)]
        L_pro = [self.label(f"p{d}") for d in range(n)]
        L_last = [self.label(f"last{d}") for d in range(n)]
        L_end = self.label("end")
        A(f"s_mov_b64 s[{S_RP}:{S_RP + 1}], %[rp]")
        A(f"s_mov_b64 s[{S_XP}:{S_XP + 1}], %[xp]")
        A(f"s_sub_u32 s{S_NS}, %[ns], 2")    # fetching half-sets behind the first: ns - 2 more borrows later (ns = 1: straight to `last`)
        # schedule words: the lap holding the chunk's first group starts 4 * phase groups = 8 * phase bytes before it
        A(f"s_lshl_b32 s{S_ACC}, %[phase], 3")
        A(f"s_mov_b64 s[{S_TP}:{S_TP + 1}], %[tp]")
        A(f"s_sub_u32 s{S_TP}, s{S_TP}, s{S_ACC}")
        A(f"s_subb_u32 s{S_TP + 1}, s{S_TP + 1}, 0")
        A(f"s_load_dwordx16 s[{TAB0}:{TAB0 + 15}], s[{S_TP}:{S_TP + 1}], 0x0")
        A(f"s_load_dwordx16 s[{TAB1}:{TAB1 + 15}], s[{S_TP}:{S_TP + 1}], 0x40")
        A(f"s_add_u32 s{S_TP}, s{S_TP}, 128")
        A(f"s_addc_u32 s{S_TP + 1}, s{S_TP + 1}, 0")
        A(f"s_mov_b32 s{S_ACC}, 0")
        A(f"s_set_gpr_idx_on s{S_ACC}, 0")   # index mode on, no operand indexed while M0[15:12] = 0
        A("s_mov_b32 m0, 0")
        for d in range(1, n):
            A(f"s_cmp_eq_u32 %[phase], {d}")
            A(f"s_cbranch_scc1 {L_pro[d]}")
        for d in range(n):
            A(f"{L_pro[d]}:")
            A(f"s_waitcnt vmcnt({2 * (n - 1)})")
            self.addrs(d, 0)
            for j in range(8):
                A(f"ds_read_b128 v[{W[j]}:{W[j] + 3}], v{AD[j]}")
            A("s_waitcnt lgkmcnt(0)")        # the schedule words (and the first reads) have landed
            A(f"s_cmp_eq_u32 %[ns], 1")
            A(f"s_cbranch_scc1 {L_last[d]}")
            A(f"s_branch {L_body[d]}")
        for d in range(n):
            self.body(d, L_body, L_last)
        for d in range(n):
            self.last(d, L_last, L_end)
        A(f"{L_end}:")
        A("s_waitcnt lgkmcnt(0)")            # a schedule load may still be in flight: its registers are not ours past this block
        A("s_mov_b32 m0, 0")
        A("s_set_gpr_idx_off")
        A(f"s_mov_b64 %[rp], s[{S_RP}:{S_RP + 1}]")
        A(f"s_mov_b64 %[xp], s[{S_XP}:{S_XP + 1}]")


def main():
    out = ["// generated by gen_acc_tiled.py -- do not edit", "#pragma once"]
    g = Gen()
    g.chunk()
    out.append(f"#define ACC_TILED_CHUNK_ASM \\\n{g.text()}")
    out.append("")
    g = Gen()
    g.A(f"s_mov_b64 s[{S_RP}:{S_RP + 1}], %[rp]")
    g.A(f"s_mov_b64 s[{S_XP}:{S_XP + 1}], %[xp]")
    for i in range(4):
        g.refill(i)
    g.A(f"s_mov_b64 %[rp], s[{S_RP}:{S_RP + 1}]")
    g.A(f"s_mov_b64 %[xp], s[{S_XP}:{S_XP + 1}]")
    out.append(f"#define ACC_TILED_RING_FILL_ASM \\\n{g.text()}")
    out.append("")
    g = Gen()
    for c in range(128):
        g.A(f"v_mov_b32 v{ACC + c}, 0")
    out.append(f"#define ACC_TILED_ZERO_ASM \\\n{g.text()}")
    out.append("")
    clob = [f'"s{r}"' for r in range(S_T, S_Q + 8)] + ['"memory"', '"scc"']
    out.append("#define ACC_TILED_CLOBBERS " + ", ".join(clob))
    # ---- four columns per LDS instruction (ranks up to 32)
    out.append("")
    g = GenQuad()
    g.chunk()
    out.append(f"#define ACC_TILED4_CHUNK_ASM \\\n{g.text()}")
    out.append("")
    g = GenQuad()
    g.A(f"s_mov_b64 s[{S_RP}:{S_RP + 1}], %[rp]")
    g.A(f"s_mov_b64 s[{S_XP}:{S_XP + 1}], %[xp]")
    for i in range(GenQuad.NS):
        g.refill(i)
    g.A(f"s_mov_b64 %[rp], s[{S_RP}:{S_RP + 1}]")
    g.A(f"s_mov_b64 %[xp], s[{S_XP}:{S_XP + 1}]")
    out.append(f"#define ACC_TILED4_RING_FILL_ASM \\\n{g.text()}")
    # ---- the pair layout on the same loop structure (half-set ring slots, no set preparation)
    out.append("")
    g = GenPairRing()
    g.chunk()
    out.append(f"#define ACC_TILED2R_CHUNK_ASM \\\n{g.text()}")
    out.append("")
    g = GenPairRing()
    g.A(f"s_mov_b64 s[{S_RP}:{S_RP + 1}], %[rp]")
    g.A(f"s_mov_b64 s[{S_XP}:{S_XP + 1}], %[xp]")
    for i in range(GenPairRing.NS):
        g.refill(i)
    g.A(f"s_mov_b64 %[rp], s[{S_RP}:{S_RP + 1}]")
    g.A(f"s_mov_b64 %[xp], s[{S_XP}:{S_XP + 1}]")
    out.append(f"#define ACC_TILED2R_RING_FILL_ASM \\\n{g.text()}")
    # ---- schedule-table bookkeeping (default): pairs and quads
    # laps of L2 prefetch ahead of the ring.  Default 0: measured at config 3 (round 5, one box, two runs each, rhs_h / rhs_w ms per
    # pass) 0 laps 10.25 / 10.21 and 10.28 / 10.24, 2 laps 10.47 / 10.47 and 10.56 / 10.48, 4 laps 10.65 / 10.66 and 10.73 / 10.65;
    # config 2 0.220 -> 0.262: the waves are not waiting for the stream, and the two extra loads per lap cost what loads cost.
    pf = int(os.environ.get("SGL_GEN_PF", "0"))
    for tag, pair in (("2T", True), ("4T", False)):
        out.append("")
        g = GenTab(pair, pf)
        g.chunk()
        out.append(f"#define ACC_TILED{tag}_CHUNK_ASM \\\n{g.text()}")
    tclob = [f'"s{r}"' for r in range(TAB0, S_Q + 8)] + ['"memory"', '"scc"'] + (['"v63"'] if pf else [])
    out.append("#define ACC_TILEDT_CLOBBERS " + ", ".join(tclob))
    out.append(f"#define ACC_TILED_PF_LAPS {pf}")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
