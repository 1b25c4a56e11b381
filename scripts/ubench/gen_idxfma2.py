#!/usr/bin/env python3
"""Generates idxfma2_gen.inc: second micro-benchmark of the row-shared accumulate, with a realistic
stream path: per wave 6 packets of 64 entry slots in flight (x 512 B + row offsets 64 B + packed
column bytes 64 B per packet, all through vector loads), x replicated over the four 16-lane rows
with v_permlane32_swap / v_permlane16_swap, column bytes moved to the scalar side with v_readlane,
one ds_read_b64 of the factor row per G = 4 slots, FP64 FMA into an M0-indexed accumulator.
Switches: lds (0: no LDS reads, f = 1), idx (0: never change the index)."""
import sys

D = 6
BASE = 128


def regs(b, n=2):
    return f"v[{b}:{b + n - 1}]"


def gen(lds, idx, nthreads=512):
    L = []
    A = L.append
    XB = [8 + 2 * d for d in range(D)]
    RB = [20 + d for d in range(D)]
    CB = [26 + d for d in range(D)]
    XQ = [32 + 2 * q for q in range(4)]
    FB = [[40 + (b * 4 + g) * 2 for g in range(4)] for b in range(2)]
    AD = [56 + g for g in range(4)]
    T0, T1 = 60, 61
    S_T = [24, 25, 26, 27]
    S_ZERO, S_CNT = 98, 99
    S_XP, S_RP, S_CP = 22, 34, 36
    LANE8, XOFF, ROFF, COFF = "%[lane8]", "%[xoff]", "%[roff]", "%[coff]"
    name = f"idx2_lds{lds}_idx{idx}_t{nthreads}"

    def issue_loads(d):
        A(f"global_load_dwordx2 {regs(XB[d])}, {XOFF}, s[{S_XP}:{S_XP + 1}]")
        A(f"global_load_dword v{RB[d]}, {ROFF}, s[{S_RP}:{S_RP + 1}]")
        A(f"global_load_dword v{CB[d]}, {COFF}, s[{S_CP}:{S_CP + 1}]")
        A(f"s_add_u32 s{S_XP}, s{S_XP}, 512")
        A(f"s_addc_u32 s{S_XP + 1}, s{S_XP + 1}, 0")
        A(f"s_add_u32 s{S_RP}, s{S_RP}, 64")
        A(f"s_addc_u32 s{S_RP + 1}, s{S_RP + 1}, 0")
        A(f"s_add_u32 s{S_CP}, s{S_CP}, 64")
        A(f"s_addc_u32 s{S_CP + 1}, s{S_CP + 1}, 0")

    def prefetch_f(rbuf, q, fbuf):
        if not lds:
            return
        for g in range(4):
            A(f"v_add_u32_dpp v{AD[g]}, v{rbuf}, {LANE8} row_newbcast:{4 * q + g} row_mask:0xf bank_mask:0xf")
        for g in range(4):
            A(f"ds_read_b64 {regs(FB[fbuf][g])}, v{AD[g]}")

    def replicate2(srcs, outs2):
        """each dword src = [R0 R1 R2 R3] (rows of 16 lanes) -> outs[q] = [Rq Rq Rq Rq]; hazard: a VALU write
        must be two wait states ahead of a permlane swap reading it"""
        for src, (o0, o1, o2, o3) in zip(srcs, outs2):
            A(f"v_mov_b32 v{o2}, v{src}")
            A(f"v_mov_b32 v{o0}, v{src}")
        A("s_nop 1")
        for src, (o0, o1, o2, o3) in zip(srcs, outs2):
            A(f"v_permlane32_swap_b32 v{o0}, v{o2}")     # o0 = [R0 R1 R0 R1], o2 = [R2 R3 R2 R3]
        A("s_nop 1")
        for src, (o0, o1, o2, o3) in zip(srcs, outs2):
            A(f"v_mov_b32 v{o1}, v{o0}")
            A(f"v_mov_b32 v{o3}, v{o2}")
        A("s_nop 1")
        for src, (o0, o1, o2, o3) in zip(srcs, outs2):
            A(f"v_permlane16_swap_b32 v{o0}, v{o1}")     # o0 = [R0 x4], o1 = [R1 x4]
            A(f"v_permlane16_swap_b32 v{o2}, v{o3}")
        A("s_nop 1")

    def packet(d, fb0):
        """consume packet in buffers d; f of its quarter 0 is in FB[fb0] (prefetched by the caller)"""
        dn = (d + 1) % D
        A(f"s_set_gpr_idx_idx s{S_ZERO}")
        A(f"s_waitcnt vmcnt({3 * (D - 1)})")
        replicate2([XB[d], XB[d] + 1], [[XQ[q] for q in range(4)], [XQ[q] + 1 for q in range(4)]])
        fb = fb0
        for q in range(4):
            if q > 0:
                A(f"s_set_gpr_idx_idx s{S_ZERO}")
            if q < 3:
                prefetch_f(RB[d], q + 1, fb ^ 1)
            else:
                A(f"s_waitcnt vmcnt({3 * (D - 2)})")  # next packet landed
                prefetch_f(RB[dn], 0, fb ^ 1)
            for g in range(4):
                A(f"v_readlane_b32 s{S_T[g]}, v{CB[d]}, {4 * q + g}")
            if q == 0:
                A("s_nop 1")
            for g in range(4):
                for i in range(4):
                    if idx:
                        A(f"s_set_gpr_idx_idx s{S_T[g]}")
                    f = FB[fb][g] if lds else FB[0][0]
                    A(f"v_fmac_f64_dpp {regs(BASE)}, {regs(XQ[q])}, {regs(f)} row_newbcast:{4 * g + i} row_mask:0xf bank_mask:0xf")
                    if idx and i < 3:
                        A(f"s_lshr_b32 s{S_T[g]}, s{S_T[g]}, 8")
            if q == 3:
                issue_loads(d)
            A("s_waitcnt lgkmcnt(0)")
            fb ^= 1
        return fb

    A(f"s_mov_b32 s{S_ZERO}, 0")
    A(f"s_mov_b32 s{S_CNT}, %[iters]")
    A(f"s_mov_b64 s[{S_XP}:{S_XP + 1}], %[xp]")
    A(f"s_mov_b64 s[{S_RP}:{S_RP + 1}], %[rp]")
    A(f"s_mov_b64 s[{S_CP}:{S_CP + 1}], %[cp]")
    for c in range(128):
        A(f"v_mov_b32 v{BASE + c}, 0")
    for d in range(D):
        issue_loads(d)
    A(f"s_waitcnt vmcnt({3 * (D - 1)})")
    if lds:
        A("s_nop 1")
        prefetch_f(RB[0], 0, 0)
    else:
        A(f"v_mov_b32 v{FB[0][0]}, 0")
        A(f"v_mov_b32 v{FB[0][0] + 1}, 0x3ff00000")
    A("s_waitcnt lgkmcnt(0)")
    A(f"s_set_gpr_idx_on s{S_ZERO}, gpr_idx(DST)")
    A("1:")
    fb = 0
    for d in range(D):
        fb = packet(d, fb)
    assert fb == 0
    A(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    A(f"s_cmp_lg_u32 s{S_CNT}, 0")
    A("s_cbranch_scc1 1b")
    A("s_set_gpr_idx_off")
    A("s_waitcnt vmcnt(0) lgkmcnt(0)")
    for c in range(64):
        A(f"global_store_dwordx2 %[ooff], {regs(BASE + 2 * c)}, %[outp]")
        A(f"v_add_u32 %[ooff], 512, %[ooff]")
    A("s_waitcnt vmcnt(0)")
    clob = [f"v{r}" for r in range(8, BASE + 128)] + [f"s{r}" for r in range(22, 100)] + ["memory", "scc", "vcc"]
    body = "\n".join(f'        "{ins}\\n\\t"' for ins in L)
    clobs = ", ".join(f'"{c}"' for c in clob)
    src = f"""
__global__ __launch_bounds__({nthreads}) __attribute__((amdgpu_num_vgpr(128))) void {name}(
    const double* __restrict__ F, const double* __restrict__ xs, const uint32_t* __restrict__ rs, const uint32_t* __restrict__ cs,
    double* __restrict__ out, int iters, long npk_wave) {{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    for (int e = threadIdx.x; e < 256 * 64; e += blockDim.x) tile[e] = F[e];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lane8 = lds0 + lane * 8;
    const unsigned xoff = lane * 8, roff = (lane & 15) * 4, coff = (lane & 15) * 4;
    const double* xp = uni(xs + wave * npk_wave * 64);
    const uint32_t* rp = uni(rs + wave * npk_wave * 16);
    const uint32_t* cp = uni(cs + wave * npk_wave * 16);
    double* outp = uni(out + wave * 64 * 64);
    const int it = __builtin_amdgcn_readfirstlane(iters);
    unsigned ooff = lane * 8;
    asm volatile(
{body}
        : [ooff] "+v"(ooff)
        : [iters] "s"(it), [xp] "s"(xp), [rp] "s"(rp), [cp] "s"(cp), [outp] "s"(outp), [lane8] "v"(lane8), [xoff] "v"(xoff), [roff] "v"(roff), [coff] "v"(coff)
        : {clobs});
}}
"""
    return name, nthreads, src


def main():
    out = ["// generated by gen_idxfma2.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <stdint.h>",
           "template <class T> __device__ __forceinline__ T* uni(T* p) { uintptr_t u = (uintptr_t)p; unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32)); return (T*)(((uintptr_t)hi << 32) | lo); }"]
    table = []
    for lds, idx in ((1, 1), (0, 1), (1, 0), (0, 0)):
        for nt in (512, 256):
            name, nthreads, src = gen(lds, idx, nt)
            out.append(src)
            table.append((name, nthreads, lds, idx))
    out.append("struct Variant { const char* name; int threads; int lds; int idx; void (*fn)(const double*, const double*, const uint32_t*, const uint32_t*, double*, int, long); };")
    out.append("static const Variant VARIANTS[] = {")
    for name, nt, lds, idx in table:
        out.append(f'    {{"{name}", {nt}, {lds}, {idx}, {name}}},')
    out.append("};")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
