// Micro-benchmark 2 of the row-shared accumulate (kernels from gen_idxfma2.py): realistic stream
// path (distinct per-wave streams, 6 packets of 64 slots in flight), verified against the host.
#include "idxfma2_gen.inc"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(r_), __LINE__); exit(1); } } while (0)
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100;  // loop iterations of 6 packets of 64 slots
    const long npk = (long)iters * 6, npk_alloc = npk + 8;
    std::vector<double> F(256 * 64);
    for (auto& v : F) v = (rnd() % 1000) / 1000.0 + 0.001;
    double* dF; CK(hipMalloc(&dF, F.size() * 8)); CK(hipMemcpy(dF, F.data(), F.size() * 8, hipMemcpyHostToDevice));
    for (const Variant& V : VARIANTS) {
        const int wpw = V.threads / 64, nwg = 256, nwaves = nwg * wpw;
        std::vector<double> xs((size_t)npk_alloc * 64, 0.0);
        std::vector<uint32_t> rs((size_t)npk_alloc * 16, 0), cs((size_t)npk_alloc * 16, 0);
        std::vector<double> ref(4096, 0.0);
        for (long p = 0; p < npk; ++p)
            for (int g = 0; g < 16; ++g) {
                const int row = rnd() % 256;
                rs[p * 16 + g] = row * 512;
                for (int i = 0; i < 4; ++i) {
                    const int c = V.idx ? rnd() % 64 : 0;
                    const double x = (rnd() % 2000) / 1000.0 - 0.5;
                    xs[p * 64 + 4 * g + i] = x;
                    cs[p * 16 + g] |= (uint32_t)(2 * c) << (8 * i);
                    for (int l = 0; l < 64; ++l) ref[c * 64 + l] = std::fma(x, V.lds ? F[row * 64 + l] : 1.0, ref[c * 64 + l]);
                }
            }
        const size_t xb = xs.size() * 8, mb = rs.size() * 4;
        char *dx, *dr, *dc; double* dout;
        CK(hipMalloc(&dx, xb * nwaves)); CK(hipMalloc(&dr, mb * nwaves)); CK(hipMalloc(&dc, mb * nwaves)); CK(hipMalloc(&dout, (size_t)nwaves * 4096 * 8));
        CK(hipMemcpy(dx, xs.data(), xb, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, rs.data(), mb, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, cs.data(), mb, hipMemcpyHostToDevice));
        for (size_t have = 1; have < (size_t)nwaves; have *= 2) {
            const size_t n = std::min(have, (size_t)nwaves - have);
            CK(hipMemcpy(dx + have * xb, dx, n * xb, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(dr + have * mb, dr, n * mb, hipMemcpyDeviceToDevice));
            CK(hipMemcpy(dc + have * mb, dc, n * mb, hipMemcpyDeviceToDevice));
        }
        CK(hipMemset(dout, 0xff, (size_t)nwaves * 4096 * 8));
        CK(hipFuncSetAttribute((const void*)V.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        V.fn<<<nwg, V.threads, 131072>>>(dF, (const double*)dx, (const uint32_t*)dr, (const uint32_t*)dc, dout, iters, npk_alloc);
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            V.fn<<<nwg, V.threads, 131072>>>(dF, (const double*)dx, (const uint32_t*)dr, (const uint32_t*)dc, dout, iters, npk_alloc);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        double maxerr = 0; long bad = 0;
        std::vector<double> got(4096);
        for (long w : {0L, (long)nwaves / 2 + 3, (long)nwaves - 1}) {
            CK(hipMemcpy(got.data(), dout + (size_t)w * 4096, 4096 * 8, hipMemcpyDeviceToHost));
            for (int e = 0; e < 4096; ++e) {
                const double d = fabs(got[e] - ref[e]);
                if (!(d <= 1e-9 * (1 + fabs(ref[e])))) ++bad;
                if (d > maxerr) maxerr = d;
            }
        }
        const double slots = (double)npk * 64 * nwaves;
        const double ns = best * 1e6 / ((double)npk * 64 * wpw);
        printf("%-24s waves/CU=%2d %8.3f ms %6.3f ns/slot/CU %5.2f cyc/slot/SIMD@2.4GHz  stream %5.2f TB/s  cfg3 (1.5e9 slots): %6.2f ms  %s (bad=%ld maxerr=%.1e)\n",
               V.name, wpw, best, ns, ns * 2.4 * 4, slots * 10.0 / best * 1e-9, 1.5e9 / slots * best, bad ? "WRONG" : "ok", bad, maxerr);
        fflush(stdout);
        CK(hipFree(dx)); CK(hipFree(dr)); CK(hipFree(dc)); CK(hipFree(dout));
    }
    return 0;
}
