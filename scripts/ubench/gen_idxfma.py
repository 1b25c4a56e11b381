#!/usr/bin/env python3
"""Generates idxfma.hip: micro-benchmark of the row-shared accumulate formulation.

One wave = 64 lanes = 64 factor rows of one 64-column block.  The 64 x FP64 column accumulators
live in VGPRs [BASE, BASE+128) and are addressed per entry through VGPR index mode
(s_set_gpr_idx_idx -> M0[7:0], DST relative).  A "packet" is 16 entry slots; G consecutive slots
share one factor row, fetched from LDS with ONE ds_read_b64 (lane = factor).  The value x of slot j
sits in lane j of every 16-lane row of a VGPR pair and reaches the FMA through DPP row_newbcast:j.

Variants (kernel name idx_<S>_g<G>_b<BASE>):
  S = smem : column indices (one dword per slot) and row offsets through s_load (triple buffered)
  S = vmem : four column bytes per dword through a vector load + v_readlane, shifts on the SALU
  G = 1, 2, 4 : slots per LDS read;  G = 16 : no LDS reads at all (VALU/SALU ceiling)
  BASE = 128 (256 VGPRs, 2 waves per SIMD) or 40 (168 VGPRs, 3 waves per SIMD)
Every kernel's result is checked against a host computation, so the timing belongs to a correct
instruction sequence (index semantics of v_fmac_f64_dpp, hazards).
"""
import sys

ROW_BYTES = 512  # 64 factors x 8 B


def regs(base, n=2):
    return f"v[{base}:{base + n - 1}]"


def gen_kernel(S, G, BASE, mode="DST", setop="idx"):
    noLDS = G == 16
    NG = 1 if noLDS else 16 // G  # groups (LDS reads) per packet
    NGL = {1: 16, 2: 8, 4: 4, 16: 4}[G]  # roff dwords stored per packet record
    XD = 3  # x buffers in flight
    # register map (all hard-coded, below BASE)
    V_LANE8, V_XOFF, V_MOFF = "%[lane8]", "%[xoff]", "%[moff]"  # compiler-allocated, below v8
    xb = [8 + 2 * i for i in range(XD)]            # v8..v13
    mb = [14 + i for i in range(XD)]               # v14..v16 (vmem variant: meta dwords)
    FB = 18
    fb = [[FB + (b * NG + g) * 2 for g in range(NG)] for b in range(2)]
    AD = FB + 4 * NG
    ad = [AD + g for g in range(NG)]
    top = AD + NG
    assert top <= BASE, (S, G, BASE, top)
    # sgpr map
    crec = 16 + NGL
    cb = [36 + i * 20 for i in range(3)] if NGL == 4 else None
    if S == "smem":
        assert NGL == 4 or BASE == 128
        # c buffers: 16 dwords each; r buffers NGL dwords each
        if NGL == 4:
            cbuf = [36, 56, 76]
            rbuf = [52, 72, 92]
        else:  # bigger roff vectors: 2-deep only does not work; use 3 x (16 + NGL) with NGL <= 8
            assert NGL == 8
            cbuf = [24, 48, 72]
            rbuf = [40, 64, 88]
    S_PTR, S_ZERO, S_CNT, S_XP = 96, 98, 99, 22
    S_MP = 34  # vmem variant: meta pointer s[34:35]
    S_T = [24 + g for g in range(4)]  # vmem variant: cpack scratch
    L = []
    A = L.append
    name = f"idx_{S}_g{G}_b{BASE}_{mode.replace(',', '')}_{setop}"
    nthreads = 512 if BASE == 128 else 768

    def fma(slot, xbuf, freg):
        return (f"v_fmac_f64_dpp {regs(BASE)}, {regs(xbuf)}, {regs(freg)} row_newbcast:{slot} row_mask:0xf bank_mask:0xf")

    def packet(i, first=False):
        """code of packet i (mod 6): consumes c/x/f of packet i, prefetches i+1 (LDS) and i+2 / i+3 (streams)"""
        c_i, c_n1, c_n2 = i % 3, (i + 1) % 3, (i + 2) % 3
        x_i = xb[i % XD]
        f_i, f_n = fb[i % 2], fb[(i + 1) % 2]
        A(f"s_set_gpr_idx_idx s{S_ZERO}")
        if S == "smem":
            if not noLDS:
                for g in range(NG):
                    A(f"v_add_u32 v{ad[g]}, s{rbuf[c_n1] + g}, {V_LANE8}")
                for g in range(NG):
                    A(f"ds_read_b64 {regs(f_n[g])}, v{ad[g]}")
            A(f"s_load_dwordx16 s[{cbuf[c_n2]}:{cbuf[c_n2] + 15}], s[{S_PTR}:{S_PTR + 1}], 0x0")
            A(f"s_load_dwordx{NGL} s[{rbuf[c_n2]}:{rbuf[c_n2] + NGL - 1}], s[{S_PTR}:{S_PTR + 1}], 0x40")
            A(f"s_add_u32 s{S_PTR}, s{S_PTR}, {crec * 4}")
            A(f"s_addc_u32 s{S_PTR + 1}, s{S_PTR + 1}, 0")
            A(f"s_waitcnt vmcnt({XD - 1})")
            for j in range(16):
                if setop == "idx":
                    A(f"s_set_gpr_idx_idx s{cbuf[c_i] + j}")
                elif setop == "m0":
                    A(f"s_mov_b32 m0, s{cbuf[c_i] + j}")
                elif setop == "every2" and j % 2 == 0:
                    A(f"s_set_gpr_idx_idx s{cbuf[c_i] + j}")
                elif setop == "every4" and j % 4 == 0:
                    A(f"s_set_gpr_idx_idx s{cbuf[c_i] + j}")
                elif setop == "nop":
                    A(f"s_and_b32 s{S_CNT + 0}, s{S_CNT}, s{S_CNT}")  # an unrelated SALU op in the same slot
                A(fma(j, x_i, f_i[0 if noLDS else j // G]))
            A(f"global_load_dwordx2 {regs(x_i)}, {V_XOFF}, s[{S_XP}:{S_XP + 1}]")
            A(f"s_add_u32 s{S_XP}, s{S_XP}, 128")
            A(f"s_addc_u32 s{S_XP + 1}, s{S_XP + 1}, 0")
            A("s_waitcnt lgkmcnt(0)")
        else:
            m_i, m_n1 = mb[i % XD], mb[(i + 1) % XD]
            # outstanding at this point: x(i+1) m(i+1) x(i+2) m(i+2); need m(i+1)
            A("s_waitcnt vmcnt(2)")
            if not noLDS:
                for g in range(NG):
                    A(f"v_add_u32_dpp v{ad[g]}, v{m_n1}, {V_LANE8} row_newbcast:{4 + g} row_mask:0xf bank_mask:0xf")
                for g in range(NG):
                    A(f"ds_read_b64 {regs(f_n[g])}, v{ad[g]}")
            for g in range(4):
                A(f"v_readlane_b32 s{S_T[g]}, v{m_i}, {g}")
            for j in range(16):
                g, q = j // 4, j % 4
                A(f"s_set_gpr_idx_idx s{S_T[g]}")
                A(fma(j, x_i, f_i[0 if noLDS else j // G]))
                if q < 3:
                    A(f"s_lshr_b32 s{S_T[g]}, s{S_T[g]}, 8")
            A(f"global_load_dwordx2 {regs(x_i)}, {V_XOFF}, s[{S_XP}:{S_XP + 1}]")
            A(f"global_load_dword v{m_i}, {V_MOFF}, s[{S_MP}:{S_MP + 1}]")
            A(f"s_add_u32 s{S_XP}, s{S_XP}, 128")
            A(f"s_addc_u32 s{S_XP + 1}, s{S_XP + 1}, 0")
            A(f"s_add_u32 s{S_MP}, s{S_MP}, 64")
            A(f"s_addc_u32 s{S_MP + 1}, s{S_MP + 1}, 0")
            A("s_waitcnt lgkmcnt(0)")

    # ---- prologue
    A(f"s_mov_b32 s{S_ZERO}, 0")
    A(f"s_mov_b32 s{S_CNT}, %[iters]")
    A(f"s_mov_b64 s[{S_XP}:{S_XP + 1}], %[xp]")
    for c in range(128):
        A(f"v_mov_b32 v{BASE + c}, 0")
    if S == "smem":
        A(f"s_mov_b64 s[{S_PTR}:{S_PTR + 1}], %[mp]")
        for b in range(2):  # packets 0 and 1
            A(f"s_load_dwordx16 s[{cbuf[b]}:{cbuf[b] + 15}], s[{S_PTR}:{S_PTR + 1}], 0x0")
            A(f"s_load_dwordx{NGL} s[{rbuf[b]}:{rbuf[b] + NGL - 1}], s[{S_PTR}:{S_PTR + 1}], 0x40")
            A(f"s_add_u32 s{S_PTR}, s{S_PTR}, {crec * 4}")
            A(f"s_addc_u32 s{S_PTR + 1}, s{S_PTR + 1}, 0")
        A("s_waitcnt lgkmcnt(0)")
        if not noLDS:
            for g in range(NG):
                A(f"v_add_u32 v{ad[g]}, s{rbuf[0] + g}, {V_LANE8}")
            for g in range(NG):
                A(f"ds_read_b64 {regs(fb[0][g])}, v{ad[g]}")
    else:
        A(f"s_mov_b64 s[{S_MP}:{S_MP + 1}], %[mp]")
    for b in range(XD):
        A(f"global_load_dwordx2 {regs(xb[b])}, {V_XOFF}, s[{S_XP}:{S_XP + 1}]")
        if S == "vmem":
            A(f"global_load_dword v{mb[b]}, {V_MOFF}, s[{S_MP}:{S_MP + 1}]")
            A(f"s_add_u32 s{S_MP}, s{S_MP}, 64")
            A(f"s_addc_u32 s{S_MP + 1}, s{S_MP + 1}, 0")
        A(f"s_add_u32 s{S_XP}, s{S_XP}, 128")
        A(f"s_addc_u32 s{S_XP + 1}, s{S_XP + 1}, 0")
    if S == "vmem" and not noLDS:
        A("s_waitcnt vmcnt(4)")  # m(0) landed
        A("s_nop 1")
        for g in range(NG):
            A(f"v_add_u32_dpp v{ad[g]}, v{mb[0]}, {V_LANE8} row_newbcast:{4 + g} row_mask:0xf bank_mask:0xf")
        for g in range(NG):
            A(f"ds_read_b64 {regs(fb[0][g])}, v{ad[g]}")
    if noLDS:
        for b in range(2):
            A(f"v_mov_b32 v{fb[b][0]}, 0")
            A(f"v_mov_b32 v{fb[b][0] + 1}, 0x3ff00000")  # f = 1.0
    A("s_waitcnt lgkmcnt(0)")
    A(f"s_set_gpr_idx_on s{S_ZERO}, gpr_idx({mode})")
    A("1:")
    for i in range(6):
        packet(i)
    A(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    A(f"s_cmp_lg_u32 s{S_CNT}, 0")
    A("s_cbranch_scc1 1b")
    A("s_set_gpr_idx_off")
    A("s_waitcnt vmcnt(0) lgkmcnt(0)")
    # ---- epilogue: store the 64 column accumulators: out[(wave * 64 + c) * 64 + lane]
    for c in range(64):
        A(f"global_store_dwordx2 %[ooff], {regs(BASE + 2 * c)}, %[outp]")
        A(f"v_add_u32 %[ooff], 512, %[ooff]")
    A("s_waitcnt vmcnt(0)")
    clob = [f"v{r}" for r in range(8, BASE + 128)] + [f"s{r}" for r in range(22, 100)] + ["memory", "scc", "vcc"]
    body = "\n".join(f'        "{ins}\\n\\t"' for ins in L)
    clobs = ", ".join(f'"{c}"' for c in clob)
    src = f"""
__global__ __launch_bounds__({nthreads}) __attribute__((amdgpu_num_vgpr({min(BASE, 128)}))) void {name}(
    const double* __restrict__ F, const double* __restrict__ xs, const uint32_t* __restrict__ meta, double* __restrict__ out,
    int iters, long xs_wave_bytes, long meta_wave_bytes) {{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    for (int e = threadIdx.x; e < 256 * 64; e += blockDim.x) tile[e] = F[e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lane8 = lds0 + lane * 8;
    const unsigned xoff = (lane & 15) * 8, moff = (lane & 15) * 4;
    const char* xp = reinterpret_cast<const char*>(xs) + wave * xs_wave_bytes;
    const char* mp = reinterpret_cast<const char*>(meta) + wave * meta_wave_bytes;
    xp = uni(xp);
    mp = uni(mp);
    double* outp = out + wave * 64 * 64;
    outp = uni(outp);
    const int it = __builtin_amdgcn_readfirstlane(iters);
    unsigned ooff = lane * 8;
    asm volatile(
{body}
        : [ooff] "+v"(ooff)
        : [iters] "s"(it), [xp] "s"(xp), [mp] "s"(mp), [outp] "s"(outp), [lane8] "v"(lane8), [xoff] "v"(xoff), [moff] "v"(moff)
        : {clobs});
}}
"""
    return name, nthreads, src, crec


def main():
    variants = []
    for S in ("smem", "vmem"):
        for G in (16, 4):
            for BASE in (128, 40):
                variants.append((S, G, BASE, "DST"))
    variants.append(("smem", 2, 128, "DST"))
    for setop in ("m0", "every2", "every4", "none", "nop"):
        variants.append(("smem", 16, 128, "DST", setop))
        variants.append(("smem", 16, 40, "DST", setop))
    out = ["// generated by gen_idxfma.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <stdint.h>",
           "template <class T> __device__ __forceinline__ T* uni(T* p) { uintptr_t u = (uintptr_t)p; unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32)); return (T*)(((uintptr_t)hi << 32) | lo); }"]
    table = []
    for v in variants:
        name, nt, src, crec = gen_kernel(*v)
        out.append(src)
        table.append((name, nt, v[0], v[1], crec, v[4] if len(v) > 4 else "idx"))
    out.append("struct Variant { const char* name; int threads; int smem_stream; int G; int crec; int setop; void (*fn)(const double*, const double*, const uint32_t*, double*, int, long, long); };")
    out.append("static const Variant VARIANTS[] = {")
    code = {"idx": 0, "m0": 1, "every2": 2, "every4": 4, "none": 99, "nop": 99}
    for name, nt, S, G, crec, setop in table:
        out.append(f'    {{"{name}", {nt}, {1 if S == "smem" else 0}, {G}, {crec}, {code[setop]}, {name}}},')
    out.append("};")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
