// Micro-benchmark: do LDS reads (ds_read_b128, DPP-computed addresses) and FP64 DPP FMAs of
// different waves overlap on a CU?  One "round" per wave = 8 address adds + 8 ds_read_b128 of the next
// octet, then 16 v_fmac_f64_dpp on the previous octet's data (the tiled accumulate's inner pattern).
// MODE 0: both, 1: only the LDS part, 2: only the FMA part.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const d2 lds_cd2;

#ifdef PLAIN_ADD
template <int J> __device__ __forceinline__ unsigned dpp_addr(unsigned r, unsigned l) { unsigned a; asm("v_add_u32 %0, %1, %2" : "=v"(a) : "v"(r + J * 416), "v"(l)); return a; }
#else
template <int J> __device__ __forceinline__ unsigned dpp_addr(unsigned r, unsigned l) {
    unsigned a; asm("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(r), "v"(l), "n"(J)); return a;
}
#endif
#ifdef PLAIN_FMA
template <int J> __device__ __forceinline__ void dpp_fmac(double& acc, double x, double w) { asm("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "v"(x), "v"(w)); }
#else
template <int J> __device__ __forceinline__ void dpp_fmac(double& acc, double x, double w) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w), "n"(J));
}
#endif
#define RD(T, J) const unsigned ad##T##J = dpp_addr<J>(ro, lane16); d2 w##T##J = *(lds_cd2*)(uintptr_t)ad##T##J;
#define FM(T, J) dpp_fmac<J>(a0, x, w##T##J.x); dpp_fmac<J>(a1, x, w##T##J.y);

template <int MODE>
__global__ __launch_bounds__(1024) void kern(double* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* t = (double*)smem;
    for (int e = threadIdx.x; e < 16384; e += blockDim.x) t[e] = e * 0.5;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned lane16 = base + (lane & 31) * 16;
    unsigned ro = ((threadIdx.x * 400) & 0xfff0);
    double a0 = 0, a1 = 0, x = 1.0 + lane;
    d2 z = {1.0, 2.0};
    d2 wp0 = z, wp1 = z, wp2 = z, wp3 = z, wp4 = z, wp5 = z, wp6 = z, wp7 = z;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 2) {
            RD(q, 0) RD(q, 1) RD(q, 2) RD(q, 3) RD(q, 4) RD(q, 5) RD(q, 6) RD(q, 7)
            if (MODE != 1) {
                dpp_fmac<0>(a0, x, wp0.x); dpp_fmac<0>(a1, x, wp0.y); dpp_fmac<1>(a0, x, wp1.x); dpp_fmac<1>(a1, x, wp1.y);
                dpp_fmac<2>(a0, x, wp2.x); dpp_fmac<2>(a1, x, wp2.y); dpp_fmac<3>(a0, x, wp3.x); dpp_fmac<3>(a1, x, wp3.y);
                dpp_fmac<4>(a0, x, wp4.x); dpp_fmac<4>(a1, x, wp4.y); dpp_fmac<5>(a0, x, wp5.x); dpp_fmac<5>(a1, x, wp5.y);
                dpp_fmac<6>(a0, x, wp6.x); dpp_fmac<6>(a1, x, wp6.y); dpp_fmac<7>(a0, x, wp7.x); dpp_fmac<7>(a1, x, wp7.y);
            } else {
                unsigned xx = __double2loint(wp0.x) ^ __double2loint(wp1.x) ^ __double2loint(wp2.x) ^ __double2loint(wp3.x) ^
                              __double2loint(wp4.x) ^ __double2loint(wp5.x) ^ __double2loint(wp6.x) ^ __double2loint(wp7.x) ^
                              __double2hiint(wp0.y) ^ __double2hiint(wp1.y) ^ __double2hiint(wp2.y) ^ __double2hiint(wp3.y) ^
                              __double2hiint(wp4.y) ^ __double2hiint(wp5.y) ^ __double2hiint(wp6.y) ^ __double2hiint(wp7.y);
                ro ^= (xx & 16);
            }
            wp0 = wq0; wp1 = wq1; wp2 = wq2; wp3 = wq3; wp4 = wq4; wp5 = wq5; wp6 = wq6; wp7 = wq7;
            ro = (ro + 400) & 0xfff0;
        } else {
            dpp_fmac<0>(a0, x, wp0.x); dpp_fmac<0>(a1, x, wp0.y); dpp_fmac<1>(a0, x, wp1.x); dpp_fmac<1>(a1, x, wp1.y);
            dpp_fmac<2>(a0, x, wp2.x); dpp_fmac<2>(a1, x, wp2.y); dpp_fmac<3>(a0, x, wp3.x); dpp_fmac<3>(a1, x, wp3.y);
            dpp_fmac<4>(a0, x, wp4.x); dpp_fmac<4>(a1, x, wp4.y); dpp_fmac<5>(a0, x, wp5.x); dpp_fmac<5>(a1, x, wp5.y);
            dpp_fmac<6>(a0, x, wp6.x); dpp_fmac<6>(a1, x, wp6.y); dpp_fmac<7>(a0, x, wp7.x); dpp_fmac<7>(a1, x, wp7.y);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + wp0.x + wp3.y;
}

template <int MODE>
void run(const char* name, int threads) {
    double* out; hipMalloc(&out, 256 * 1024 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)kern<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    kern<MODE><<<256, threads, 140000>>>(out, 100);
    hipEventRecord(e0);
    kern<MODE><<<256, threads, 140000>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s waves/CU=%2d: %.3f ms, %.1f ns per round (8 reads + 16 FMAs per wave, all waves)\n", name, threads / 64, ms, ms * 1e6 / iters);
    hipFree(out);
}
int main() {
    for (int th : {256, 512, 1024}) { run<1>("lds only", th); run<2>("fma only", th); run<0>("both", th); }
    return 0;
}
