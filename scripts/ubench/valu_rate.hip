// Micro-benchmark: issue rate of the VALU instructions the tiled accumulate is made of
// (v_fmac_f64, v_fmac_f64_dpp row_newbcast, v_add_u32_dpp, v_permlane16_swap), per SIMD,
// with W waves per SIMD.  Independent accumulators so that only issue rate is measured.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void kern(double* out, int iters, double xin) {
    double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double x = xin + threadIdx.x, w = 1.0 + 1e-9 * threadIdx.x;
    unsigned u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3, u4 = 4, u5 = 5, u6 = 6, u7 = 7, r = threadIdx.x * 8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MODE == 0) {
                asm volatile("v_fmac_f64 %0, %8, %9\n v_fmac_f64 %1, %8, %9\n v_fmac_f64 %2, %8, %9\n v_fmac_f64 %3, %8, %9\n"
                             "v_fmac_f64 %4, %8, %9\n v_fmac_f64 %5, %8, %9\n v_fmac_f64 %6, %8, %9\n v_fmac_f64 %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
            } else if (MODE == 1) {
                asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
            } else if (MODE == 2) {
                asm volatile("v_add_u32_dpp %0, %8, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %1, %8, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %2, %8, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %3, %8, %3 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %4, %8, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %5, %8, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %6, %8, %6 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %7, %8, %7 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(r));
            } else if (MODE == 3) {
                asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
                             "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(r));
            } else if (MODE == 4) {
                asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                             "v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n v_permlane16_swap_b32 %4, %6\n v_permlane16_swap_b32 %5, %7"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));
            } else if (MODE == 5) {  // fma with an SGPR-pair multiplicand (no DPP)
                asm volatile("v_fmac_f64 %0, s[8:9], %8\n v_fmac_f64 %1, s[8:9], %8\n v_fmac_f64 %2, s[8:9], %8\n v_fmac_f64 %3, s[8:9], %8\n"
                             "v_fmac_f64 %4, s[8:9], %8\n v_fmac_f64 %5, s[8:9], %8\n v_fmac_f64 %6, s[8:9], %8\n v_fmac_f64 %7, s[8:9], %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
            } else if (MODE == 6) {  // v_pk_fma_f32-free check: v_fma_f64 VOP3
                asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                             "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));
            } else if (MODE == 8) {  // round 6: v_cvt_f64_f32 (what an f32 copy of the factor in LDS would cost per factor)
                float f = (float)w;
                asm volatile("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %8\n v_cvt_f64_f32 %2, %8\n v_cvt_f64_f32 %3, %8\n"
                             "v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %8\n v_cvt_f64_f32 %6, %8\n v_cvt_f64_f32 %7, %8"
                             : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(f));
            } else if (MODE == 7) {  // v_readlane pairs
                unsigned s0, s1, s2, s3;
                asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %6, 7\n v_readlane_b32 %3, %7, 9\n"
                             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(u0), "v"(u1), "v"(u2), "v"(u3));
                u4 += s0 ^ s1 ^ s2 ^ s3;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
}

template <int MODE>
void run(const char* name, int threads, int per_iter) {
    double* out; hipMalloc(&out, 256 * 1024 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<MODE><<<256, threads>>>(out, 100, 0.5);
    hipEventRecord(e0);
    kern<MODE><<<256, threads>>>(out, iters, 0.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 64 / 4.0;
    const double instr_per_simd = (double)iters * per_iter * waves_per_simd;
    printf("%-22s waves/SIMD=%.0f: %.3f ms, %.3f ns per wave-instr per SIMD (%.2f clk @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}
int main() {
    for (int th : {256, 512, 1024}) {
        run<0>("v_fmac_f64", th, 32);
        run<6>("v_fma_f64 (VOP3)", th, 32);
        run<5>("v_fmac_f64 sgpr", th, 32);
        run<1>("v_fmac_f64_dpp", th, 32);
        run<3>("v_add_u32", th, 32);
        run<2>("v_add_u32_dpp", th, 32);
        run<4>("v_permlane16_swap", th, 32);
        run<7>("v_readlane_b32", th, 16);
        run<8>("v_cvt_f64_f32", th, 32);
    }
    return 0;
}
