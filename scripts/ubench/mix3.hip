// Micro-benchmark 3: do ds_read_b128 and FP64 FMAs overlap on a SIMD, and does the DPP form of the FMA matter?
// Half-iterations with hand-placed registers (gen_mix3.py): the 8 reads of one buffer are in flight while the 16
// FMAs consume the other, then lgkmcnt(0), then the roles swap -- no compiler copies between the halves.
// Modes: 1 reads only, 2 dpp FMAs only, 3 reads + dpp FMAs, 4 plain FMAs only, 5 reads + plain FMAs,
//        6 reads + plain FMAs with an SGPR multiplicand, 7 dpp address adds + reads + dpp FMAs (the kernel's octet,
//        16 independent accumulators), 8 / 9 the same with 2 / 4 accumulator chains (the kernel has 2 per wave).
// All modes read one row per half-wave (conflict-free).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "mix3_gen.inc"

#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory"
template <int MODE>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void kern(double* out, int iters) {
    asm volatile("" ::: "v255");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* t = (double*)smem;
    for (int e = threadIdx.x; e < 16384; e += blockDim.x) t[e] = 1.0 + e * 1e-6;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lane16 = base + (lane & 31) * 16;
    const unsigned ro = ((((lane & 15) * 53 + (threadIdx.x >> 6) * 7 + (lane >> 5) * 29) % 340) * 400);   // like the kernel: every entry pair its own two rows (one per half-wave), conflict-free
    const double x = 1.0 + 1e-9 * lane;
    asm volatile(MIX3_INIT ::: "memory");
    asm volatile("v_add_u32 v40, %0, %1\n v_add_u32 v41, 400, v40\n v_add_u32 v42, 800, v40\n v_add_u32 v43, 1200, v40\n"
                 "v_add_u32 v44, 1600, v40\n v_add_u32 v45, 2000, v40\n v_add_u32 v46, 2400, v40\n v_add_u32 v47, 2800, v40\n"
                 "s_mov_b64 s[40:41], 1.0" :: "v"(lane16), "v"(ro) : "s40", "s41", "memory");
    unsigned goff = lane * 8;
    const uintptr_t gpu_ = (MODE >= 15) ? (uintptr_t)((char*)out + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (4u << 20))
                                         : (uintptr_t)(out + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64);   // a wave-private line to re-load
    const double* gp = (const double*)(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(gpu_ >> 32)) << 32) |
                                       (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)gpu_));
    asm volatile("s_mov_b32 s42, 0x7fffffff\n s_mov_b32 s43, 0x8000" ::: "s42", "s43");
    if (MODE >= 11) asm volatile("s_mov_b32 m0, 0\n s_set_gpr_idx_on m0, 0\n s_mov_b32 m0, 0" ::: "memory");
    const int iters2 = (MODE == 14) ? iters / 16 : (MODE >= 10) ? iters / 2 : iters;   // mode 14: eight sets of code per iteration (12 KB)   // a set = four octets = two "rounds" of the other modes
    for (int it = 0; it < iters2; ++it) {
        if (MODE == 1) asm volatile(MIX3_BODY_1 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16) : "memory");
        if (MODE == 2) asm volatile(MIX3_BODY_2 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16) : "memory");
        if (MODE == 3) asm volatile(MIX3_BODY_3 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16) : "memory");
        if (MODE == 4) asm volatile(MIX3_BODY_4 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16) : "memory");
        if (MODE == 5) asm volatile(MIX3_BODY_5 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16) : "memory");
        if (MODE == 6) asm volatile(MIX3_BODY_6 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16) : "memory");
        if (MODE == 7) asm volatile(MIX3_BODY_7 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16) : "memory");
        if (MODE == 8) asm volatile(MIX3_BODY_8 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16) : "memory");
        if (MODE == 9) asm volatile(MIX3_BODY_9 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16) : "memory");
        if (MODE == 10) asm volatile(MIX3_BODY_10 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16), [goff] "v"(goff), [gp] "s"(gp) : "memory", "scc", "s42", "s43", "v32", "v33", "v34", "v35", "v36", "v37", "v38");
        if (MODE == 16) asm volatile(MIX3_BODY_16 : [goff] "+v"(goff) : [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16), [gp] "s"(gp) : "memory", "scc", "s42", "s43", "v32", "v33", "v34", "v35", "v36", "v37", "v38");
        if (MODE == 17) asm volatile(MIX3_BODY_17 : [goff] "+v"(goff) : [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16), [gp] "s"(gp) : "memory", "scc", "s42", "s43", "v32", "v33", "v34", "v35", "v36", "v37", "v38");
        if (MODE == 15) asm volatile(MIX3_BODY_15 : [goff] "+v"(goff) : [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16), [gp] "s"(gp) : "memory", "scc", "s42", "s43", "v32", "v33", "v34", "v35", "v36", "v37", "v38");
        if (MODE == 14) asm volatile(MIX3_BODY_13 MIX3_BODY_13 MIX3_BODY_13 MIX3_BODY_13 MIX3_BODY_13 MIX3_BODY_13 MIX3_BODY_13 MIX3_BODY_13 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16), [goff] "v"(goff), [gp] "s"(gp) : "memory", "scc", "s42", "s43", "v32", "v33", "v34", "v35", "v36", "v37", "v38");
        if (MODE == 13) asm volatile(MIX3_BODY_13 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16), [goff] "v"(goff), [gp] "s"(gp) : "memory", "scc", "s42", "s43", "v32", "v33", "v34", "v35", "v36", "v37", "v38");
        if (MODE == 12) asm volatile(MIX3_BODY_12 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16), [goff] "v"(goff), [gp] "s"(gp) : "memory", "scc", "s42", "s43", "v32", "v33", "v34", "v35", "v36", "v37", "v38");
        if (MODE == 11) asm volatile(MIX3_BODY_11 :: [x] "v"(x), [ro] "v"(ro), [lane16] "v"(lane16), [goff] "v"(goff), [gp] "s"(gp) : "memory", "scc", "s42", "s43", "v32", "v33", "v34", "v35", "v36", "v37", "v38");
    }
    if (MODE >= 11) asm volatile("s_mov_b32 m0, 0\n s_set_gpr_idx_off" ::: "memory");
    double s;
    asm volatile("v_add_f64 %0, v[112:113], v[114:115]\n v_add_f64 %0, %0, v[140:141]\n v_add_f64 %0, %0, v[48:49]" : "=v"(s));
    if (MODE < 15) out[blockIdx.x * blockDim.x + threadIdx.x] = s; else if (s == 12345.678) out[0] = s;
}

template <int MODE>
void run(const char* name, int threads, int iters = 10000) {
    double* out; hipMalloc(&out, MODE >= 15 ? ((size_t)256 * 16 * (4u << 20) + (1u << 20)) : (size_t)256 * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)kern<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    kern<MODE><<<256, threads, 140000>>>(out, 100);
    hipEventRecord(e0);
    kern<MODE><<<256, threads, 140000>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/CU=%2d: %7.1f ns per octet round (8 reads / 16 FMAs per wave)\n", name, threads / 64, ms * 1e6 / iters / 2);
    hipFree(out);
}
int main(int argc, char** argv) {
    if (argc > 1) {   // sustained run of the kernel-like modes: does the rate hold over ~1 s (clocks under load)?
        const int it = atoi(argv[1]);
        for (int rep = 0; rep < 3; ++rep) { run<7>("7 sustained", 512, it); run<11>("11 sustained", 512, it); }
        return 0;
    }
    for (int th : {256, 512}) {
        run<1>("1 reads only", th); run<2>("2 dpp FMAs only", th); run<3>("3 reads + dpp FMAs", th); run<4>("4 plain FMAs only", th);
        run<5>("5 reads + plain FMAs", th); run<6>("6 reads + sgpr-x FMAs", th); run<7>("7 dpp adds + reads + dpp FMAs", th);
        run<8>("8 = 7 with 2 accumulator chains", th); run<9>("9 = 7 with 4 accumulator chains", th);
        run<10>("10 whole set, hand-scheduled", th); run<11>("11 = 10, accumulators through M0", th);
        run<12>("12 = 11, another pair every set", th); run<13>("13 the kernel: one buffer, reads behind FMAs", th); run<14>("14 = 13, eight copies of the set (12 KB of code)", th);
        run<15>("15 = 13 + a real 768 B / set stream from HBM", th); run<16>("16 = 15 with 8 sets in flight", th); run<17>("17 = 15 with 16 sets in flight", th);
    }
    return 0;
}
