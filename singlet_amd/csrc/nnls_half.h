// nnls_half_kernel<KH>: nnls (src/singlet.cpp:229-250) for 64 < k <= 104 with TWO LANES PER COLUMN.
//
// The lane-per-column kernel (nnls_lane.h) keeps b[k] and x[k] of its column in one lane's registers; above k = 64 that
// is more than a lane has (4 k VGPRs), so its k > 64 instances hold x in a global scratch and still spill 1 - 4 KB per lane:
// 3 x off the k^2 scaling of the k <= 64 instances.  Here lane c (c < 32) and lane 32 + c of a wave share column c of the
// wave's 32 columns: the lower half-wave holds coordinates 0 .. KH - 1 of b and x, the upper half KH .. 2 KH - 1 -- everything
// in registers, no scratch.  Per coordinate i: b_i and x_i reach both halves with v_permlane32_swap (a swap of two copies
// leaves the lower half's value in both halves of one register and the upper half's in the other), both halves run the
// same step (tol stays consistent), and each lane updates its KH entries of b.  The Gram is staged in LDS as in nnls_lane.h;
// a lane reads the 16-entry pieces of ITS half of row i, so the DPP row broadcast of FMA jj delivers G[i, jj] to the lower
// half and G[i, KH + jj] to the upper.  Same operations in the same order per column: bit-identical to the oracle's sweeps.
// Re-packing passes as in nnls_lane.h (NnlsPass), 32 columns per wave.
#pragma once
#include "sgl_internal.h"
#include "nnls_static_for.h"
#ifndef SGL_NNLS_REPACK_NUM
#define SGL_NNLS_REPACK_NUM 3
#define SGL_NNLS_REPACK_DEN 8
#endif

#ifndef SGL_HALF_G2
#define SGL_HALF_G2 1
#endif
template <int J>
__device__ __forceinline__ void half_dpp_fmac(double& acc, double g, double nd) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(g), "v"(nd), "n"(J));
}
// lo = the lower half-wave's v in all 64 lanes, hi = the upper half-wave's
__device__ __forceinline__ void half_bcast(double v, double& lo, double& hi) {
    // (the builtin, not inline asm: the compiler then inserts the wait states the swap needs after a VALU write of its operands)
    const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    const int a0 = (int)r0[0], b0 = (int)r0[1], a1 = (int)r1[0], b1 = (int)r1[1];
    lo = __hiloint2double(a1, a0);
    hi = __hiloint2double(b1, b0);
}

// KH > 52 (k = 105 ... 128): b alone takes 2 KH <= 128 VGPRs of the wave's 256 at two waves per SIMD; x then lives in the
// AGPR half of the register file -- named registers a[2 j : 2 j + 1], outside hipcc's allocation (it has no other use for
// AGPRs in this kernel; tests/test_kernel_codegen.py holds it to that) -- at one wave per SIMD (512 registers).  A
// coordinate reads and writes its x once: two v_accvgpr_read + two v_accvgpr_write per step.
template <int J>
__device__ __forceinline__ double half_xa_read() {
    int lo, hi;
    asm volatile("v_accvgpr_read_b32 %0, a[%2]\n\tv_accvgpr_read_b32 %1, a[%3]\n\ts_nop 1" : "=v"(lo), "=v"(hi) : "n"(2 * J), "n"(2 * J + 1));
    return __hiloint2double(hi, lo);
}
template <int J>
__device__ __forceinline__ void half_xa_write(double v) {
    asm volatile("v_accvgpr_write_b32 a[%2], %0\n\tv_accvgpr_write_b32 a[%3], %1" ::"v"(__double2loint(v)), "v"(__double2hiint(v)), "n"(2 * J), "n"(2 * J + 1));
}
template <int KH>
__device__ __forceinline__ void half_xa_reserve() {
    if constexpr (KH <= 56) asm volatile("" ::: "a111");
    else if constexpr (KH <= 60) asm volatile("" ::: "a119");
    else asm volatile("" ::: "a127");
    static_assert(KH <= 64, "x in AGPRs: at most 64 coordinates per lane");
}

template <int KH>
constexpr size_t nnls_half_lds_bytes() {
    constexpr int KP = 2 * KH, GS = ((KP + 15) / 16) * 16;
    return ((size_t)(KP + 1) * GS + 32 + 2 * (size_t)KP) * sizeof(double);
}

template <int KH>
__global__ __launch_bounds__(KH > 52 ? 256 : 512) __attribute__((amdgpu_waves_per_eu(KH > 52 ? 1 : 2))) void nnls_half_kernel(
    const double* __restrict__ Gpad, double* __restrict__ B, double* __restrict__ X, const int64_t* __restrict__ col_nnz, int k,
    int64_t ncols, double L1, double L2, unsigned long long* __restrict__ sweep_counter, NnlsPass ps) {
    constexpr int KP = 2 * KH, NG = (KP + 15) / 16, GS = NG * 16, NGH = (KH + 15) / 16;
    constexpr bool G2 = SGL_HALF_G2 != 0 && KH <= 48;   // Gram rows one coordinate ahead (registers permitting)
    constexpr bool XA = KH > 52;                        // x in named AGPRs (one wave per SIMD)
    if constexpr (XA) half_xa_reserve<KH>();
    const int64_t n_in = ps.list ? (int64_t)*ps.count : ncols;
    const int cpb = (int)(blockDim.x >> 1);   // columns per workgroup
    if ((int64_t)blockIdx.x * cpb >= n_in) return;
    extern __shared__ __attribute__((aligned(16))) double nnls_half_lds[];
    double* const Gl = nnls_half_lds;                         // (KP + 1) x GS (+ 32 doubles of slack behind the last row)
    double* const Dl = nnls_half_lds + (KP + 1) * GS + 32;    // (G_jj, 1 / G_jj)
    for (int e = threadIdx.x; e < (KP + 1) * GS + 32; e += blockDim.x) Gl[e] = e < (KP + 1) * GS ? Gpad[e] : 0.0;
    for (int j = threadIdx.x; j < KP; j += blockDim.x) {
        Dl[2 * j] = Gpad[j * GS + j];
        Dl[2 * j + 1] = Gpad[KP * GS + j];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int64_t gid = (int64_t)blockIdx.x * cpb + (threadIdx.x >> 6) * 32 + (lane & 31);   // position in this pass
    const bool in_range = gid < n_in;
    const int64_t col = in_range ? (ps.list ? (int64_t)ps.list[gid] : gid) : 0;
    const bool resume = ps.list != nullptr && !ps.fresh;
    const bool valid = in_range && (resume || col_nnz == nullptr || col_nnz[col] != 0);
    const bool to_end = (ps.next_list == nullptr) || n_in <= (int64_t)ps.final_below;
    constexpr int KLOW = KP - 7;   // coordinates below it always exist (an instance serves KP - 7 <= k <= KP)
    int one = 1;
    double b[KH], x[XA ? 1 : KH];
    double* bp = B + col * k + half * KH;
    double* xp = X + col * k + half * KH;
    const int kh = half ? k - KH : KH;   // coordinates this lane holds (k >= KP - 7 > KH)
    static_for<KH>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        b[j] = (valid && j < kh) ? bp[j] : 0.0;
        const double x0 = (valid && j < kh) ? xp[j] : 0.0;
        if constexpr (XA) half_xa_write<j>(x0); else x[j] = x0;
    });
    const double kd = (double)k;
    double tol = 1.0;
    int it = 0;
    if (valid && resume) {
        tol = ps.tol_state[col];
        it = (int)ps.it_state[col];
    }
    int gofs = 0, ran = 0;
    const int n_act0 = __popcll(__ballot(valid && it < 100 && (tol / kd) > 1e-8) & 0xffffffffull);
    const int gbase = half * KH + (lane & 15);   // this lane's piece of a Gram row
    while (true) {
        const bool go = valid && it < 100 && (tol / kd) > 1e-8;
        const int n_act = __popcll(__ballot(go) & 0xffffffffull);
        if (n_act == 0) break;
        if (!to_end && n_act * SGL_NNLS_REPACK_DEN < n_act0 * SGL_NNLS_REPACK_NUM) break;  // re-pack the stragglers
        ++ran;
        if (go) tol = 0.0;
        asm volatile("" : "+s"(gofs));   // (opaque zero: keeps the LDS reads inside the sweep loop, see nnls_lane.h)
        const int g0 = gofs + gbase;
        double gn[NGH];
        if (G2) {
#pragma unroll
            for (int m = 0; m < NGH; ++m) gn[m] = Gl[g0 + 16 * m];
        }
        double dnext0 = Dl[gofs], dnext1 = Dl[gofs + 1];
        static_for<KP>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int owner = i >= KH ? 1 : 0, ii = i - owner * KH;
            bool run_i = i < k;
            if (i < KLOW) { asm volatile("" : "+s"(one)); run_i = one != 0; }   // opaque, always true: one basic block per coordinate
            if (run_i) {
                double grow[NGH];
                if (G2) {
#pragma unroll
                    for (int m = 0; m < NGH; ++m) grow[m] = gn[m];
                    if (i + 1 < KP) {   // this lane's piece of row i + 1
#pragma unroll
                        for (int m = 0; m < NGH; ++m) gn[m] = Gl[g0 + (i + 1) * GS + 16 * m];
                    }
                } else {   // read at the head of the coordinate: the step's ~40 dependent instructions cover the LDS latency
#pragma unroll
                    for (int m = 0; m < NGH; ++m) grow[m] = Gl[g0 + i * GS + 16 * m];
                }
                const double gii = dnext0, rii = dnext1;
                if (i + 1 < KP) { dnext0 = Dl[gofs + 2 * (i + 1)]; dnext1 = Dl[gofs + 2 * (i + 1) + 1]; }
                __builtin_amdgcn_sched_barrier(0);
                double blo, bhi, xlo, xhi, xown;
                if constexpr (XA) xown = half_xa_read<ii>(); else xown = x[ii];
                half_bcast(b[ii], blo, bhi);
                half_bcast(xown, xlo, xhi);
                const double bi = owner ? bhi : blo, xi = owner ? xhi : xlo;
                // b_i / G_ii, correctly rounded, from the correctly rounded reciprocal (Markstein; see nnls_lane.h)
                const double q0 = bi * rii;
                const double rem = fma(-q0, gii, bi);
                const double diff0 = fma(rem, rii, q0);
                double xv = xi;
                const double nd = sgl_nnls_step(diff0, xv, tol, go, L1, L2);
                const double xnew = (half == owner) ? xv : xown;
                if constexpr (XA) half_xa_write<ii>(xnew); else x[ii] = xnew;
                static_for<KH>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    half_dpp_fmac<(j & 15)>(b[j], grow[j >> 4], nd);
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        it += go ? 1 : 0;
    }
    const bool unfinished = valid && it < 100 && (tol / kd) > 1e-8;  // only possible when !to_end
    if (valid) {
        static_for<KH>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            double xj;
            if constexpr (XA) xj = half_xa_read<j>(); else xj = x[j];
            if (j < kh) xp[j] = xj;
        });
    } else if constexpr (XA) {
        asm volatile("" ::: "memory");
    }
    if (unfinished) {
        static_for<KH>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j < kh) bp[j] = b[j];
        });
        if (half == 0) {
            ps.tol_state[col] = tol;
            ps.it_state[col] = (uint8_t)it;
        }
    }
    const unsigned long long um = __ballot(unfinished) & 0xffffffffull;
    if (um != 0ull) {  // wave-aggregated append (the lower half-wave speaks for the columns)
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(ps.next_count, (unsigned)__popcll(um));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (unfinished && half == 0) ps.next_list[base + (unsigned)__popcll(um & ((1ull << lane) - 1ull))] = (int32_t)col;
    }
    if (sweep_counter != nullptr) {
        int s = (valid && !unfinished && half == 0) ? it : 0;  // a column's sweeps are booked once, when it stops
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0 && (s != 0 || ran != 0)) {
            atomicAdd(sweep_counter, (unsigned long long)s);
            atomicAdd(sweep_counter + 2, (unsigned long long)ran);  // sweeps this wave actually executed
        }
    }
}
