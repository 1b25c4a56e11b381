// Micro-benchmark: LDS read rate per CU for ds_read_b64 / ds_read_b128 with per-lane addresses
// lane*8 (or lane*16) + wave-uniform row offset, N reads in flight per wait, W waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) const double lds_cd;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const d2 lds_cd2;

template <int MODE, int NIF, int STR>
__global__ void kern(double* out, int iters, int stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* t = (double*)smem;
    for (int e = threadIdx.x; e < 16384; e += blockDim.x) t[e] = e * 0.5;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    double acc = 0, acc2 = 0;
    unsigned ro = (threadIdx.x >> 6) * 400;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            double w[NIF];
            const unsigned a0 = base + (ro & 0x7fff) + lane * 8;
#pragma unroll
            for (int j = 0; j < NIF; ++j) { w[j] = *(lds_cd*)(uintptr_t)(a0 + j * STR); }
            unsigned x = 0;
#pragma unroll
            for (int j = 0; j < NIF; ++j) x ^= (unsigned)__double2loint(w[j]);
            acc += (double)x;
        } else {
            d2 w[NIF];
            const unsigned a0 = base + (ro & 0x7ff0) + (lane & 31) * 16;
#pragma unroll
            for (int j = 0; j < NIF; ++j) { w[j] = *(lds_cd2*)(uintptr_t)(a0 + j * STR); }
            unsigned x = 0;
#pragma unroll
            for (int j = 0; j < NIF; ++j) x ^= (unsigned)__double2loint(w[j].x) ^ (unsigned)__double2loint(w[j].y);
            acc += (double)x;
        }
        ro += STR * NIF;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + acc2;
}

template <int MODE, int NIF, int STR>
void run(const char* name, int threads) {
    double* out; hipMalloc(&out, 256 * 1024 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)kern<MODE, NIF, STR>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
    kern<MODE, NIF, STR><<<256, threads, 140000>>>(out, 100, 400);
    hipEventRecord(e0);
    kern<MODE, NIF, STR><<<256, threads, 140000>>>(out, iters, 400);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_cu = (double)iters * NIF * (threads / 64);
    printf("%-12s stride=%d threads=%4d NIF=%2d: %.3f ms, %.2f ns per LDS wave-instr per CU (%.2f clk @2.1GHz), %s\n", name, STR, threads, NIF, ms,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1, MODE ? "2 entries/instr" : "1 entry/instr");
    hipFree(out);
}
int main() {
    run<0, 16, 400>("b64", 512); run<0, 16, 512>("b64", 512); run<0, 16, 520>("b64", 512); run<0, 16, 256>("b64", 512);
    run<1, 8, 400>("b128", 512); run<1, 8, 512>("b128", 512); run<1, 8, 528>("b128", 512); run<1, 8, 416>("b128", 512);
    run<1, 8, 400>("b128", 1024); run<1, 8, 512>("b128", 1024);
    return 0;
}
