// nnls (src/singlet.cpp:229-250): cyclic coordinate-descent NNLS, one solve
// per column, <= 100 sweeps, stop when tol / k <= 1e-8.  FP64 VALU bound.
//
// Two mappings:
//  * nnls_lane_kernel<KP>: ONE LANE PER COLUMN.  b[KP] and x[KP] live in VGPRs,
//    the sweep is fully unrolled, the shared Gram G is wave-uniform and reaches
//    the FMAs as scalar (SGPR) operands.  Every VALU lane does useful work; the
//    wave runs until its slowest column converges.  Used for k <= 64 with a
//    Gram shared by all columns (the c_nmf / c_project_model path).
//  * nnls_wave_kernel<R>: ONE WAVE PER COLUMN, lanes over the k coordinates.
//    Handles any k <= 256 and a per-column Gram (the masked path, where
//    a_i = a - asub differs per column, src/singlet.cpp:460-463).
//
// Sequential semantics are kept exactly: coordinate order, in-place b, the
// `tol = 1` overwrite on a clamp, clamp only when x_i != 0, L1 subtracted from
// every step, L2 * x_i added to every step.  The only
// arithmetic difference to an SSE2 build of the reference is FMA contraction of
// b - G*delta.
#include "sgl_internal.h"
#include <utility>
#include <type_traits>

// compile-time loop: guarantees that b[] / x[] are only ever indexed by constants
// (so they live in VGPRs) regardless of the optimiser's unroll thresholds.
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int KP>
__global__ __launch_bounds__(256) void nnls_lane_kernel(const double* __restrict__ Gpad, const double* __restrict__ B,
                                                        double* __restrict__ X, const int64_t* __restrict__ col_nnz,
                                                        int k, int64_t ncols, double L1, double L2,
                                                        unsigned long long* __restrict__ sweep_counter) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // empty columns are skipped and keep their stale values (src/singlet.cpp:340)
    const bool valid = (col < ncols) && (col_nnz == nullptr || col_nnz[col] != 0);
    double b[KP], x[KP];
    const double* bp = B + col * k;
    double* xp = X + col * k;
    static_for<KP>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        b[j] = (valid && j < k) ? bp[j] : 0.0;
        x[j] = (valid && j < k) ? xp[j] : 0.0;
    });
    const double kd = (double)k;
    double tol = 1.0;
    int it = 0;
    int gofs = 0;
    while (true) {
        const bool go = valid && it < 100 && (tol / kd) > 1e-8;
        if (!__any(go)) break;
        if (go) tol = 0.0;
        // launder a (wave-uniform, always zero) offset once per sweep: the k*k scalar loads of the
        // Gram must be re-issued every sweep instead of being hoisted out of the loop and spilled.
        // The pointer itself keeps its provenance (global, read-only) so the loads stay s_load.
        asm volatile("" : "+s"(gofs));
        const double* __restrict__ Gs = Gpad + gofs;
        static_for<KP>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (i < k) {
                const double xi = x[i];
                double diff = b[i] / Gs[i + KP * i];
                diff -= L1;                 // exact no-op when L1 == 0
                diff = fma(L2, xi, diff);   // exact no-op when L2 == 0 (x >= 0)
                const bool clamp = -diff > xi;
                const bool c2 = clamp && (xi != 0.0);
                const bool upd = (!clamp) && (diff != 0.0);
                const double xn = c2 ? 0.0 : (upd ? xi + diff : xi);
                double delta = c2 ? -xi : (upd ? diff : 0.0);
                delta = go ? delta : 0.0;
                x[i] = go ? xn : xi;
                const double tadd = fabs(diff / (xn + 1e-15));
                const double tnew = c2 ? 1.0 : (upd ? tol + tadd : tol);
                tol = go ? tnew : tol;
                const double nd = -delta;
                static_for<KP>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    b[j] = fma(Gs[j + KP * i], nd, b[j]);
                });
            }
        });
        it += go ? 1 : 0;
    }
    if (valid) {
        static_for<KP>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j < k) xp[j] = x[j];
        });
    }
    if (sweep_counter != nullptr) {
        int s = it, mx = it;
        for (int off = 32; off > 0; off >>= 1) {
            s += __shfl_down(s, off, 64);
            mx = max(mx, __shfl_down(mx, off, 64));
        }
        if ((threadIdx.x & 63) == 0 && s != 0) {
            atomicAdd(sweep_counter, (unsigned long long)s);
            atomicAdd(sweep_counter + 2, (unsigned long long)mx);  // diagnostic: sweeps the wave actually ran
        }
    }
}

int k_nnls_lane(hipStream_t s, const double* Gpad, int KP, const double* B, double* X, const int64_t* col_nnz, int k,
                int64_t ncols, double L1, double L2, unsigned long long* sweep_counter) {
    if (ncols <= 0) return SGL_OK;
    dim3 g((unsigned)((ncols + 255) / 256)), b(256);
#define SGL_NNLS(K_) case K_: nnls_lane_kernel<K_><<<g, b, 0, s>>>(Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter); break
    switch (KP) {
        SGL_NNLS(4); SGL_NNLS(8); SGL_NNLS(12); SGL_NNLS(16); SGL_NNLS(20); SGL_NNLS(24); SGL_NNLS(28); SGL_NNLS(32);
        SGL_NNLS(36); SGL_NNLS(40); SGL_NNLS(44); SGL_NNLS(48); SGL_NNLS(52); SGL_NNLS(56); SGL_NNLS(60); SGL_NNLS(64);
        default: sgl_set_error("k_nnls_lane: unsupported KP=%d", KP); return SGL_EINVAL;
    }
#undef SGL_NNLS
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// ---------------------------------------------------------------------------
__device__ __forceinline__ double rl64(double v, int lane) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffull), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int R>
__global__ __launch_bounds__(256) void nnls_wave_kernel(const double* __restrict__ G, int64_t gstride,
                                                        const double* __restrict__ B, double* __restrict__ X,
                                                        const int64_t* __restrict__ col_nnz, int k, int64_t ncols,
                                                        double L1, double L2, unsigned long long* __restrict__ sweep_counter) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const double kd = (double)k;
    int total_sweeps = 0;
    for (int64_t col = wave; col < ncols; col += nwaves) {
        if (col_nnz != nullptr && col_nnz[col] == 0) continue;
        const double* Gc = G + col * gstride;
        double b[R], x[R], gd[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = lane + 64 * r;
            b[r] = (j < k) ? B[col * k + j] : 0.0;
            x[r] = (j < k) ? X[col * k + j] : 0.0;
            gd[r] = (j < k) ? Gc[(int64_t)j * k + j] : 1.0;
        }
        double tol = 1.0;
        int it = 0;
        for (; it < 100 && (tol / kd) > 1e-8; ++it) {
            tol = 0.0;
            for (int i = 0; i < k; ++i) {
                const int ir = i >> 6, il = i & 63;
                double bsel = b[0], xsel = x[0], gsel = gd[0];
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    if (ir == r) { bsel = b[r]; xsel = x[r]; gsel = gd[r]; }
                }
                const double bi = rl64(bsel, il), xi = rl64(xsel, il), gii = rl64(gsel, il);
                double diff = bi / gii;
                diff -= L1;
                diff = fma(L2, xi, diff);
                double delta = 0.0, xn = xi;
                if (-diff > xi) {
                    if (xi != 0.0) { delta = -xi; tol = 1.0; xn = 0.0; }
                } else if (diff != 0.0) {
                    xn = xi + diff;
                    delta = diff;
                    tol += fabs(diff / (xn + 1e-15));
                }
                if (delta != 0.0 || xn != xi) {  // wave-uniform
                    const double nd = -delta;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int j = lane + 64 * r;
                        if (j < k) b[r] = fma(Gc[(int64_t)i * k + j], nd, b[r]);
                        if (ir == r && il == lane) x[r] = xn;
                    }
                }
            }
        }
        total_sweeps += it;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = lane + 64 * r;
            if (j < k) X[col * k + j] = x[r];
        }
    }
    if (sweep_counter != nullptr && lane == 0 && total_sweeps != 0) {
        atomicAdd(sweep_counter, (unsigned long long)total_sweeps);
        atomicAdd(sweep_counter + 2, (unsigned long long)total_sweeps);
    }
}

int k_nnls_wave(hipStream_t s, const double* G, int64_t gstride, const double* B, double* X, const int64_t* col_nnz,
                int k, int64_t ncols, double L1, double L2, unsigned long long* sweep_counter) {
    if (ncols <= 0) return SGL_OK;
    if (k > SGL_MAX_K) { sgl_set_error("k_nnls_wave: k=%d > %d", k, SGL_MAX_K); return SGL_EINVAL; }
    int64_t blocks = (ncols + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 g((unsigned)blocks), b(256);
    const int R = (k + 63) / 64;
    switch (R) {
        case 1: nnls_wave_kernel<1><<<g, b, 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter); break;
        case 2: nnls_wave_kernel<2><<<g, b, 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter); break;
        case 3: nnls_wave_kernel<3><<<g, b, 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter); break;
        default: nnls_wave_kernel<4><<<g, b, 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter); break;
    }
    HIPCHK(hipGetLastError());
    return SGL_OK;
}
