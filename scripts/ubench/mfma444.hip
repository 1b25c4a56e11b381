// Operand layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by probing: A one-hot at lane la, B one-hot at lane lb,
// every (la, lb) pair in its own wave; prints for each pair the lanes of D that become non-zero.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* out) {
    const int lane = threadIdx.x & 63, w = blockIdx.x;   // w = la * 64 + lb
    const int la = w >> 6, lb = w & 63;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? (double)(lb + 2) : 0.0;
    double d = 0.0;
    d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, 0, 0, 0);
    out[(size_t)w * 64 + lane] = d;
}
int main() {
    double* d;
    hipMalloc(&d, sizeof(double) * 4096 * 64);
    probe<<<4096, 64>>>(d);
    std::vector<double> h(4096 * 64);
    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // for each la: which lb give a non-zero D, and where
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" (B%d->D%d)", lb, l);
        printf("\n");
    }
    return 0;
}
