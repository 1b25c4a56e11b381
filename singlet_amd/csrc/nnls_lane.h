// nnls_lane_kernel<KP>: nnls (src/singlet.cpp:229-250) with ONE LANE PER COLUMN (see kernels_nnls.hip
// for the two mappings).  Included by kernels_nnls_lane{1,2}.hip, which instantiate disjoint sets of
// KP so that the (long) compiles run in parallel.
//
// Re-packing.  Lanes of a wave run in lock-step, so a wave is busy until its slowest column stops:
// at config 3 the columns need 31 sweeps on average but a wave runs 46.  The solve is therefore
// done in PASSES: a wave leaves a pass as soon as fewer than 3/8 of the lanes it started with are
// still iterating, writes the state of the unfinished columns back (b in place in B, x, the sweep
// count and the running tol) and appends them to a list; the next pass runs the listed columns
// densely packed.  A column's own sequence of sweeps is unchanged (same order, same arithmetic, same
// stop test after every sweep), so results are bit-identical to the one-pass kernel whatever the
// packing; only the order in which columns land in the list varies from run to run.
#pragma once
#include "sgl_internal.h"
// a wave leaves a pass when fewer than NUM / DEN of the lanes it started with are still iterating
#ifndef SGL_NNLS_WPE
#define SGL_NNLS_WPE 2   // minimum waves per SIMD the register allocation must allow
#endif
#ifndef SGL_NNLS_GRAM_LDS
#define SGL_NNLS_GRAM_LDS 1
#endif
#ifndef SGL_NNLS_REPACK_NUM
#define SGL_NNLS_REPACK_NUM 3
#define SGL_NNLS_REPACK_DEN 8
#endif

#include "nnls_static_for.h"

// GV = false: the Gram reaches the FMAs as scalar operands (s_load; row stride KP).
// GV = true : row i of the Gram is fetched with ceil(KP / 16) coalesced vector loads, every 16-lane row of
//   the wave holding the same 16 entries (row stride GS = KP rounded up to 16), and entry j reaches FMA j as a
//   DPP row broadcast (v_fmac_f64_dpp ... row_newbcast:j%16).  Vector loads return in order and cost no
//   SGPRs: a whole row no longer has to fit the ~100 free SGPRs (the cliff above k = 50) and hipcc can
//   keep the next rows in flight.
template <int J>
__device__ __forceinline__ void nnls_dpp_fmac(double& acc, double g, double nd) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(g), "v"(nd), "n"(J));
}
template <int J>
__device__ __forceinline__ double nnls_dpp_bcast(double g) {
    int lo = __double2loint(g), hi = __double2hiint(g), rl, rh;
    asm("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(rl) : "v"(lo), "n"(J));
    asm("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(rh) : "v"(hi), "n"(J));
    return __hiloint2double(rh, rl);
}

// XM = true (k > 64): x does not fit the register file next to b any more (4 k VGPRs): it lives in a
// per-launch scratch xt[i * xt_stride + position] (coalesced over the lanes) and is read PF coordinates
// ahead; b stays in VGPRs.
template <int KP, bool GV, bool XM = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SGL_NNLS_WPE))) void nnls_lane_kernel(const double* __restrict__ Gpad, double* __restrict__ B,
                                                        double* __restrict__ X, const int64_t* __restrict__ col_nnz,
                                                        int k, int64_t ncols, double L1, double L2,
                                                        unsigned long long* __restrict__ sweep_counter, NnlsPass ps) {
    // columns of this pass: all of them (first pass) or the list written by the previous pass
    const int64_t n_in = ps.list ? (int64_t)*ps.count : ncols;
    if ((int64_t)blockIdx.x * blockDim.x >= n_in) return;
    // GV, k <= 64: the padded Gram (rows 0 .. KP: the Gram and the reciprocals of its diagonal, (KP + 1) x GS doubles, 33 KB
    // at KP = 64) is staged ONCE per workgroup in LDS and the sweeps read their rows from there: immediate offsets off one
    // per-lane base (no per-coordinate address arithmetic), LDS latency instead of the vector cache's.
    constexpr bool GLDS = GV && !XM && SGL_NNLS_GRAM_LDS;
    constexpr int GS_ = ((KP + 15) / 16) * 16;
    __shared__ double Gl[GLDS ? (KP + 1) * GS_ : 1];
    __shared__ __attribute__((aligned(16))) double Dl[GLDS ? 2 * KP : 2];   // (G_jj, 1 / G_jj) pairs: one uniform 16-byte read per coordinate
    if (GLDS) {
        for (int e = threadIdx.x; e < (KP + 1) * GS_; e += blockDim.x) Gl[e] = Gpad[e];
        for (int j = threadIdx.x; j < KP; j += blockDim.x) {
            Dl[2 * j] = Gpad[j * GS_ + j];
            Dl[2 * j + 1] = Gpad[KP * GS_ + j];
        }
        __syncthreads();
    }
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = gid < n_in;
    const int64_t col = in_range ? (ps.list ? (int64_t)ps.list[gid] : gid) : 0;
    // empty columns are skipped and keep their stale values (src/singlet.cpp:340)
    const bool resume = ps.list != nullptr && !ps.fresh;   // a later pass: the column's state was saved by the previous one
    const bool valid = in_range && (resume || col_nnz == nullptr || col_nnz[col] != 0);
    const bool to_end = (ps.next_list == nullptr) || n_in <= (int64_t)ps.final_below;
    constexpr int PF = XM ? 2 : 4;   // coordinates of x read ahead (XM)
    // An instance serves KP - 1 <= k <= KP (KP - 7 <= k above 64): for the coordinates below KLOW the run-time test
    // `i < k` is always true.  hipcc implemented it as a lane mask kept in (spilled) SGPRs -- ~10 instructions per
    // coordinate -- but simply dropping it makes the whole sweep ONE basic block, and then the register allocator
    // spills (368 B of scratch per lane at KP = 50 against 20, nnls_h 7.1 -> 10.7 ms at config 3).  So those
    // coordinates keep a branch, on an opaque always-true scalar: s_cmp + s_cbranch, no mask (nnls_h 6.6 -> 5.8 ms).
    constexpr int KLOW = XM ? KP - 7 : KP - 1;
    int one = 1;
    constexpr bool G2 = !XM;          // one-row-ahead double buffer of the Gram rows (registers permitting)
    double b[KP], x[XM ? PF : KP];
    double* bp = B + col * k;
    double* xp = X + col * k;
    double* __restrict__ xt = XM ? ps.xt + gid : nullptr;  // this lane's column of the scratch
    const int64_t xs = ps.xt_stride;
    static_for<KP>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        b[j] = (valid && j < k) ? bp[j] : 0.0;
        if (!XM) x[j] = (valid && j < k) ? xp[j] : 0.0;
    });
    if (XM) {
        // only lanes that own a column touch the scratch: a lane past the end of the list would land in
        // another column's slot of the next row
        if (valid)
            for (int j = 0; j < k; ++j) xt[j * xs] = xp[j];
    }
    const double kd = (double)k;
    double tol = 1.0;
    int it = 0;
    if (valid && resume) {
        tol = ps.tol_state[col];
        it = (int)ps.it_state[col];
    }
    int gofs = 0, ran = 0;
    const int n_act0 = __popcll(__ballot(valid && it < 100 && (tol / kd) > 1e-8));
    while (true) {
        const bool go = valid && it < 100 && (tol / kd) > 1e-8;
        const int n_act = __popcll(__ballot(go));
        if (n_act == 0) break;
        if (!to_end && n_act * SGL_NNLS_REPACK_DEN < n_act0 * SGL_NNLS_REPACK_NUM) break;  // re-pack the stragglers
        ++ran;
        if (go) tol = 0.0;
        // launder a (wave-uniform, always zero) offset once per sweep: the k*k scalar loads of the
        // Gram must be re-issued every sweep instead of being hoisted out of the loop and spilled.
        // The pointer itself keeps its provenance (global, read-only) so the loads stay s_load.
        asm volatile("" : "+s"(gofs));
        const double* __restrict__ Gs = Gpad + gofs;
        constexpr int NG = (KP + 15) / 16, GS = NG * 16;
        const double* __restrict__ Gvg = Gpad + gofs + (threadIdx.x & 15);
        const int gl0 = gofs + (int)(threadIdx.x & 15);
        // entry X of this lane's column of the padded Gram (LDS keeps its address space: no generic pointer)
        auto Gv = [&](int X) -> double { if constexpr (GLDS) return Gl[gl0 + X]; else return Gvg[X]; };
        // GV: explicit one-row-ahead software pipeline of the Gram rows (g2[parity]), fenced with
        // scheduling barriers: left alone, hipcc hoists the loads of dozens of rows of this straight-line
        // code and spills (kilobytes of scratch per lane at k > 64).
        // row KP of the padded Gram holds the correctly rounded reciprocals 1 / G_jj (k_pad_gram): the step
        // b_i / G_ii then costs a multiply and two FMAs instead of an 11-instruction IEEE division (below)
        // (not in the k > 64 instances: they have no registers to spare -- measured 11 % slower at k = 100 -- and
        // keep the division)
        constexpr bool RCP = !XM;
        double rrow[(GV && RCP && !GLDS) ? NG : 1];
        if (GV && RCP && !GLDS) {
#pragma unroll
            for (int m = 0; m < NG; ++m) rrow[m] = Gv(KP * GS + 16 * m);
        }
        // GLDS: the diagonal pair of the coming coordinate, read (uniform address: a broadcast) one coordinate ahead
        double dnext0 = 0.0, dnext1 = 1.0;
        if (GLDS) { dnext0 = Dl[gofs]; dnext1 = Dl[gofs + 1]; }
        double g2[G2 ? 2 : 1][NG];
        if (GV && G2) {
#pragma unroll
            for (int m = 0; m < NG; ++m) g2[0][m] = Gv(16 * m);
        }
        if (XM) {  // the first PF coordinates of this sweep (slot = coordinate % PF)
#pragma unroll
            for (int q = 0; q < PF; ++q) x[q] = valid ? xt[q * xs] : 0.0;
        }
        static_for<KP>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            bool run_i = i < k;
            if (i < KLOW) { asm volatile("" : "+s"(one)); run_i = one != 0; }   // opaque, always true: keeps one basic block per coordinate
            if (run_i) {
                const double xi = x[XM ? (i % PF) : i];
                if (XM && (i + PF < KLOW || i + PF < k)) x[i % PF] = valid ? xt[(i + PF) * xs] : 0.0;  // x of coordinate i + PF (same slot)
                double grow[NG];
                double gii, rii;
                if (GV) {
                    if (G2) {
                        if (i + 1 < KLOW || i + 1 < k) {
#pragma unroll
                            for (int m = 0; m < NG; ++m) g2[(i + 1) & 1][m] = Gv((i + 1) * GS + 16 * m);
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < NG; ++m) g2[0][m] = Gv(i * GS + 16 * m);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < NG; ++m) grow[m] = g2[G2 ? (i & 1) : 0][m];
                    if (GLDS) {
                        gii = dnext0; rii = dnext1;
                        if (i + 1 < KP) { dnext0 = Dl[gofs + 2 * (i + 1)]; dnext1 = Dl[gofs + 2 * (i + 1) + 1]; }
                    } else {
                        gii = nnls_dpp_bcast<(i & 15)>(grow[i >> 4]);
                        rii = RCP ? nnls_dpp_bcast<(i & 15)>(rrow[(RCP && !GLDS) ? (i >> 4) : 0]) : 0.0;
                    }
                } else {
                    gii = Gs[i + KP * i];
                    rii = Gs[KP * KP + i];
                }
                // b_i / G_ii, correctly rounded, from the correctly rounded reciprocal (Markstein): q = RN(b r),
                // rem = b - q G_ii exactly (FMA), RN(q + rem r).  G_ii is the same for all columns and sweeps.
                double diff0;
                if (RCP) {
                    const double q0 = b[i] * rii;
                    const double rem = fma(-q0, gii, b[i]);
                    diff0 = fma(rem, rii, q0);
                } else {
                    diff0 = sgl_div_normal(b[i], gii);
                }
                // l.235-247 through sgl_nnls_step (branch-free; a stopped column takes a zero step)
                double xv = xi;
                const double nd = sgl_nnls_step(diff0, xv, tol, go, L1, L2);
                if (XM) {
                    if (go) xt[i * xs] = xv;
                } else {
                    x[i] = xv;
                }
                static_for<KP>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (GV) nnls_dpp_fmac<(j & 15)>(b[j], grow[j >> 4], nd);
                    else b[j] = fma(Gs[j + KP * i], nd, b[j]);
                });
                if (GV) __builtin_amdgcn_sched_barrier(0);
            }
        });
        it += go ? 1 : 0;
    }
    const bool unfinished = valid && it < 100 && (tol / kd) > 1e-8;  // only possible when !to_end
    if (valid) {
        if (XM) {
            for (int j = 0; j < k; ++j) xp[j] = xt[j * xs];
        } else {
            static_for<KP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j < k) xp[j] = x[j];
            });
        }
    }
    if (unfinished) {
        static_for<KP>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j < k) bp[j] = b[j];
        });
        ps.tol_state[col] = tol;
        ps.it_state[col] = (uint8_t)it;
    }
    if (valid && !unfinished && ps.prev_it != nullptr) ps.prev_it[col] = (uint8_t)it;   // packing key of the next solve
    const unsigned long long um = __ballot(unfinished);
    if (um != 0ull) {  // wave-aggregated append
        const int lane = threadIdx.x & 63;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(ps.next_count, (unsigned)__popcll(um));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (unfinished) ps.next_list[base + (unsigned)__popcll(um & ((1ull << lane) - 1ull))] = (int32_t)col;
    }
    if (sweep_counter != nullptr) {
        int s = (valid && !unfinished) ? it : 0;  // a column's sweeps are booked once, when it stops
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0 && (s != 0 || ran != 0)) {
            atomicAdd(sweep_counter, (unsigned long long)s);
            atomicAdd(sweep_counter + 2, (unsigned long long)ran);  // sweeps this wave actually executed
        }
    }
}

// GV_ is fixed per translation unit: scalar operands up to KP = 40 (3-10 % faster there), vector loads +
// DPP broadcast from KP = 42 (11 % faster at k = 50, 2.3x at k = 56 .. 64, where a row no longer fits the
// free SGPRs).  nnls_gram_stride() in kernels_nnls.hip must agree.
#define SGL_NNLS_CASE(K_, GV_) \
    case K_: nnls_lane_kernel<K_, GV_><<<g, b, 0, s>>>(Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps); break
#define SGL_NNLS_CASE_XM(K_) \
    case K_: nnls_lane_kernel<K_, true, true><<<g, b, 0, s>>>(Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps); break
