// nnls (src/singlet.cpp:229-250) at ranks 129 - 256 against a SHARED Gram: FOUR LANES PER COLUMN (round 6).
//
// The lane-per-column solve keeps b[k] and x[k] of a column in one lane (k <= 64), the two-lane solve in two (k <= 128,
// nnls_half.h).  Here the four lanes c, 16 + c, 32 + c, 48 + c of a wave share column c of the wave's 16 columns: lane row q holds
// coordinates q KQ .. (q + 1) KQ - 1 of b and x (KQ = KP / 4 <= 64), everything in registers.  Per coordinate i: b_i and x_i reach all
// four rows from the owner row (ds_bpermute through __shfl), all rows run the same step -- tol stays consistent in the four lanes of
// a column, as in nnls_half.h -- and each lane updates its KQ entries of b with v_fmac_f64_dpp: a lane holds the entries
// q KQ + c + 16 m of Gram row i, the DPP row broadcast of FMA t delivers G[i, q KQ + t] to the whole row.  The Gram (up to 0.5 MB)
// stays in global memory / L2, row i + 1's pieces are fetched while coordinate i is worked on; (G_ii, 1 / G_ii) are staged once
// per workgroup in LDS.  One chain issue serves 16 columns where the four-columns-per-wave solve (nnls_quad_global.h, used for
// these ranks until this kernel) serves four: ~(40 + KQ) instructions per coordinate and 16 columns against ~60 per four.
// Same operations in the same order per column as nnls_wave_kernel / the oracle: the same bits (tests/test_gpu_ops.py).
// An instance KQ serves 4 KQ - 15 <= k <= 4 KQ.
#pragma once
#include "sgl_internal.h"
#include "nnls_static_for.h"

#include <algorithm>

template <int J>
__device__ __forceinline__ void quarter_dpp_fmac(double& acc, double g, double nd) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(g), "v"(nd), "n"(J));
}
__device__ __forceinline__ double quarter_from_row(double v, int src_lane) {
    const int lo = __shfl(__double2loint(v), src_lane, 64), hi = __shfl(__double2hiint(v), src_lane, 64);
    return __hiloint2double(hi, lo);
}

template <int KQ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KQ > 44 ? 1 : 2))) void nnls_quarter_kernel(
    const double* __restrict__ G, const double* __restrict__ B, double* __restrict__ X, const int64_t* __restrict__ col_nnz, int k,
    int64_t ncols, double L1, double L2, unsigned long long* __restrict__ sweep_counter, const int32_t* __restrict__ order,
    uint8_t* __restrict__ prev_it) {
    constexpr int KP = 4 * KQ, NGQ = (KQ + 15) / 16, KLOW = KP - 15;
    __shared__ __attribute__((aligned(16))) double Dl[2 * KP];   // (G_jj, 1 / G_jj)
    for (int j = threadIdx.x; j < KP; j += blockDim.x) {
        const double g = (j < k) ? G[(size_t)j * k + j] : 1.0;
        Dl[2 * j] = g;
        Dl[2 * j + 1] = 1.0 / g;   // correctly rounded reciprocal of the diagonal
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, q = lane >> 4, c16 = lane & 15;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const double kd = (double)k;
    const int kq = std::max(0, std::min(KQ, k - q * KQ));   // coordinates this lane holds
    // this lane's pieces of a Gram row: entries q KQ + c16 + 16 m, clamped into the row (an entry past the rank only ever meets a
    // b that is never read)
    int goff[NGQ];
#pragma unroll
    for (int m = 0; m < NGQ; ++m) goff[m] = std::min(q * KQ + c16 + 16 * m, k - 1);
    long long total_sweeps = 0, ran_total = 0;
    const int64_t ngroups = (ncols + 15) >> 4;
    for (int64_t grp = wave; grp < ngroups; grp += nwaves) {
        // order (optional): the columns in descending order of the sweeps their previous solve needed (kernels_nnls.hip, "packing by
        // sweep count"): the 16 columns of a wave then stop at about the same sweep.  A column's arithmetic does not depend on its place.
        const int64_t pos = grp * 16 + c16;
        const int64_t col = pos < ncols ? (order ? (int64_t)order[pos] : pos) : ncols;
        const bool valid = col < ncols && (col_nnz == nullptr || col_nnz[col] != 0);
        const double* bp = B + (valid ? col : 0) * k + q * KQ;
        double* xp = X + (valid ? col : 0) * k + q * KQ;
        double b[KQ], x[KQ];
        static_for<KQ>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const bool v = valid && j < kq;
            b[j] = v ? bp[j] : 0.0;
            x[j] = v ? xp[j] : 0.0;
        });
        double tol = 1.0;
        int it = 0, ran = 0, one = 1;
        while (true) {
            const bool go = valid && it < 100 && (tol / kd) > 1e-8;
            if (__ballot(go) == 0ull) break;
            ++ran;
            if (go) tol = 0.0;
            const double* __restrict__ gp = G;   // row i of the running coordinate
            double g2[2][NGQ];   // two row buffers by the parity of the coordinate: no copies (a VALU copy in front of a DPP read would also be a hazard)
#pragma unroll
            for (int m = 0; m < NGQ; ++m) g2[0][m] = gp[goff[m]];
            double dn0 = Dl[0], dn1 = Dl[1];
            static_for<KP>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int owner = i / KQ, ii = i - owner * KQ;
                bool run_i = i < k;
                if (i < KLOW) { asm volatile("" : "+s"(one)); run_i = one != 0; }   // opaque, always true: one basic block per coordinate
                if (run_i) {
                    constexpr int pb = i & 1;
                    gp += k;
                    if constexpr (i + 1 < KP) {   // this lane's pieces of row i + 1, in flight during this coordinate
                        const bool more = (i + 1 < KLOW) || (i + 1 < k);
                        if (more) {
#pragma unroll
                            for (int m = 0; m < NGQ; ++m) g2[pb ^ 1][m] = gp[goff[m]];
                        }
                    }
                    const double gii = dn0, rii = dn1;
                    if constexpr (i + 1 < KP) { dn0 = Dl[2 * (i + 1)]; dn1 = Dl[2 * (i + 1) + 1]; }
                    __builtin_amdgcn_sched_barrier(0);
                    const double bi = quarter_from_row(b[ii], 16 * owner + c16);
                    const double xi = quarter_from_row(x[ii], 16 * owner + c16);
                    // b_i / G_ii, correctly rounded, from the correctly rounded reciprocal (Markstein; see nnls_lane.h)
                    const double q0 = bi * rii;
                    const double rem = fma(-q0, gii, bi);
                    const double diff0 = fma(rem, rii, q0);
                    double xv = xi;
                    const double nd = sgl_nnls_step(diff0, xv, tol, go, L1, L2);
                    x[ii] = (q == owner) ? xv : x[ii];
                    static_for<KQ>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        quarter_dpp_fmac<(j & 15)>(b[j], g2[pb][j >> 4], nd);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            it += go ? 1 : 0;
        }
        if (valid) {
            static_for<KQ>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j < kq) xp[j] = x[j];
            });
            if (q == 0) {
                total_sweeps += it;
                if (prev_it != nullptr) prev_it[col] = (uint8_t)it;   // packing key of the next solve
            }
        }
        ran_total += ran;
    }
    if (sweep_counter != nullptr) {
        for (int off = 32; off > 0; off >>= 1) total_sweeps += __shfl_down(total_sweeps, off, 64);
        if (lane == 0 && (total_sweeps != 0 || ran_total != 0)) {
            atomicAdd(sweep_counter, (unsigned long long)total_sweeps);
            atomicAdd(sweep_counter + 2, (unsigned long long)ran_total);
        }
    }
}

template <int KQ>
static int launch_quarter(hipStream_t s, const double* G, const double* B, double* X, const int64_t* col_nnz, int k, int64_t ncols, double L1,
                          double L2, unsigned long long* sweep_counter, const int32_t* order, uint8_t* prev_it) {
    const int64_t ngroups = (ncols + 15) / 16;
    const int64_t wgs = std::max<int64_t>(1, std::min<int64_t>((ngroups + 3) / 4, 256 * (KQ > 44 ? 1 : 2)));
    nnls_quarter_kernel<KQ><<<dim3((unsigned)wgs), dim3(256), 0, s>>>(G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, order, prev_it);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

