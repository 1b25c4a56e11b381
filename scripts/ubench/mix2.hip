// Micro-benchmark 2: which pieces of the tiled accumulate's octet serialise on a SIMD?
//  M0: 8 dpp adds                      M1: 8 ds_read_b128 (fixed addresses)     M2: 16 dpp FMAs (2 chains)
//  M3: adds + reads                    M4: reads + FMAs (previous data)          M5: adds + reads + FMAs
//  M6: 16 FMAs, 8 chains               M7: reads (b64) + FMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

#define ADD(J) asm volatile("v_add_u32_dpp %0, %1, %2 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "=v"(ad[J]) : "v"(ro), "v"(lane16));
#define RDA(J) asm volatile("ds_read_b128 %0, %1" : "=v"(w[J]) : "v"(ad[J]));
#define RDH(J) asm volatile("ds_read_b64 %0, %1" : "=v"(w[J].x) : "v"(ad[J]));
#define FM(J) asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %2, %4 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1) : "v"(x), "v"(wp[J].x), "v"(wp[J].y));
#define FM8(J) asm volatile("v_fmac_f64_dpp %0, %2, %3 row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %2, %4 row_newbcast:" #J " row_mask:0xf bank_mask:0xf" : "+v"(c[J]), "+v"(c[J + 8]) : "v"(x), "v"(wp[J].x), "v"(wp[J].y));
#define ALL8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

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
    double c[16];
    for (int j = 0; j < 16; ++j) c[j] = j;
    d2 w[8], wp[8];
    unsigned ad[8];
    for (int j = 0; j < 8; ++j) { wp[j] = d2{1.0, 2.0}; w[j] = wp[j]; ad[j] = lane16 + ((ro + j * 400) & 0xfff0); }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 3 || MODE == 5) { ALL8(ADD) }
        if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) { ALL8(RDA) }
        if (MODE == 7) { ALL8(RDH) }
        if (MODE == 2 || MODE == 4 || MODE == 5 || MODE == 7) { ALL8(FM) }
        if (MODE == 6) { ALL8(FM8) }
        if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5 || MODE == 7) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
            for (int j = 0; j < 8; ++j) wp[j] = w[j];
        }
        if (MODE == 0) asm volatile("" :: "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]));
    }
    double s = a0 + a1 + wp[0].x + wp[3].y;
    for (int j = 0; j < 16; ++j) s += c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
    printf("%-28s waves/CU=%2d: %7.1f ns per round\n", name, threads / 64, ms * 1e6 / iters);
    hipFree(out);
}
int main() {
    for (int th : {256, 512, 1024}) {
        run<0>("M0 8 dpp adds", th); run<1>("M1 8 reads b128", th); run<2>("M2 16 FMA 2 chains", th); run<6>("M6 16 FMA 8 chains", th);
        run<3>("M3 adds+reads", th); run<4>("M4 reads+FMA", th); run<5>("M5 adds+reads+FMA", th); run<7>("M7 reads b64 + FMA", th);
    }
    return 0;
}
