// nnls_quad_global_kernel<NR>: four columns per wave, the Gram in global memory (kernels_nnls.hip: per-column Grams of the masked
// path, k <= 128; kernels_nnls_quad_big1 / 2.hip, round 6: NR = 9 .. 16 against the SHARED Gram of a plain fit at ranks 129 - 256).
#pragma once
#include "sgl_internal.h"
#include "nnls_static_for.h"
#include <algorithm>

// b_i and x_i (and 1 / g_ii) of a coordinate for the 16 lanes of every row in one statement.  (__builtin_amdgcn_update_dpp
// initialises its `old` operand -- a v_mov per dword -- although a row_newbcast has no invalid source lane; this is the bare
// v_mov_b32_dpp.  hipcc pads no hazards around inline asm: the s_nop covers the two wait states between a VALU write of
// a source -- the row update of the coordinate before -- and its DPP read.)
template <int J>
__device__ __forceinline__ void nnls_row_bcast2(double bsrc, double xsrc, double& bi, double& xi) {
    const int blo = __double2loint(bsrc), bhi = __double2hiint(bsrc), xlo = __double2loint(xsrc), xhi = __double2hiint(xsrc);
    int r0, r1, r2, r3;
    asm("s_nop 1\n\t"
        "v_mov_b32_dpp %0, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %2, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %3, %7 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(blo), "v"(bhi), "v"(xlo), "v"(xhi), "n"(J));
    bi = __hiloint2double(r1, r0);
    xi = __hiloint2double(r3, r2);
}

// The same four-columns-per-wave solve for ranks whose triangles do not fit LDS (50 < k <= 128): the column's Gram
// stays in global memory and row i + 1 is fetched (each 16-lane row its own column's 128-byte pieces) while coordinate
// i is worked on.  nnls_wave_kernel spends ~49 instructions per column and coordinate step, 45 of them uniform over the
// wave (eight v_readlane, the step's chain); here one issue of the chain serves four columns and the broadcasts are
// DPP moves: ~13 instructions per column and step.  Every row is read in every sweep (the wave kernel skips the rows
// of coordinates that do not move; here a row is skipped when its coordinate moves in none of the four columns, and
// fetched ahead only if it moved in the previous sweep).  Same arithmetic, same order: bit-identical results.
// Four waves per SIMD (128 registers: 20 - 104 B of scratch per lane at NR >= 5).  While the solve was bound by HBM (rows
// skipped per wave: 6.2 TB/s) the occupancy made no difference; with the rows skipped per column it fetches 3.7 TB/s and
// more waves in flight pay: nnls_h k = 72: 40.4 -> 36.1 ms, 100: 67.5 -> 62.3, 128: 92.0 -> 89.7 (30 000 x 200 000).
#ifndef SGL_QG_WPE
#define SGL_QG_WPE 4
#endif

template <int NR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NR <= 8 ? SGL_QG_WPE : 2))) void nnls_quad_global_kernel(const double* __restrict__ G, int64_t gstride,
                                                              const double* __restrict__ B, double* __restrict__ X,
                                                              const int64_t* __restrict__ col_nnz, int k, int64_t ncols,
                                                              double L1, double L2, unsigned long long* __restrict__ sweep_counter) {
    const int lane = threadIdx.x, grp = lane >> 4, l = lane & 15;
    const double kd = (double)k;
    long long total_sweeps = 0, ran_total = 0;
    const int64_t nquads = (ncols + 3) >> 2;
    __shared__ __attribute__((aligned(16))) double dgl[4 * 16 * NR * 2];
    for (int64_t quad = blockIdx.x; quad < nquads; quad += gridDim.x) {
        const int64_t col = quad * 4 + grp;
        const bool cvalid = col < ncols && (col_nnz == nullptr || col_nnz[col] != 0);
        const double* __restrict__ Gc = G + (cvalid ? col : 0) * gstride;
        // (G_jj, 1 / G_jj) of the four columns: in LDS, not in 4 NR registers per lane -- a coordinate needs its pair once,
        // as one 16-byte read at a constant offset a coordinate ahead, where four DPP broadcasts served it before
        double b[NR], x[NR];
        bool irr = false;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = l + 16 * r;
            const bool v = cvalid && j < k;
            b[r] = v ? B[col * k + j] : 0.0;
            x[r] = v ? X[col * k + j] : 0.0;
            const double gdj = v ? Gc[(int64_t)j * k + j] : 1.0;
            const double rdj = 1.0 / gdj;   // correctly rounded reciprocal of the diagonal
            dgl[(grp * 16 * NR + j) * 2] = gdj;
            dgl[(grp * 16 * NR + j) * 2 + 1] = rdj;
            irr = irr || !__builtin_isnormal(rdj);
        }
        const bool any_irr = __ballot(irr) != 0ull;   // a zero (or otherwise irregular) diagonal entry in one of the four columns
        __builtin_amdgcn_wave_barrier();
        const double* const dgc = dgl + (size_t)grp * 16 * NR * 2;   // this column's pairs
        double tol = 1.0;
        int it = 0, ran = 0, one = 1;
        // Row i of a column's Gram is only needed when coordinate i MOVES in that column (otherwise b += 0 * row), and the
        // kernel is bound by these reads (FETCH_SIZE: 675 GB per pass over 200 000 columns at k = 100 = 6.2 TB/s, the
        // copy rate of the part).  Which coordinates move is nearly the same from sweep to sweep, so a 16-lane row
        // fetches row i ahead only if coordinate i moved in ITS column in the previous sweep (all of them in the
        // first); a coordinate that moves without its row at hand fetches it then (rare).  The masks are per column
        // (one bit per coordinate, the same value in the 16 lanes of a column): a column that has stopped, or whose
        // coordinate rests at its bound, reads nothing while its three neighbours in the wave go on.  Rows not fetched
        // leave stale -- finite -- values in g: they meet nd = 0.
        constexpr int NW = (16 * NR + 31) / 32;
        unsigned act[NW], cur[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) act[w] = ~0u;
        // Two row buffers by the parity of the coordinate (no copies), every load of a row unconditional under ONE exec
        // mask: the lanes past the rank (only in piece NR - 1: an instance serves 16 (NR - 1) < k <= 16 NR) read the row's
        // last entry instead -- their b is never looked at.  (Written as `j < k ? gp[16 r] : 0.0` per piece, hipcc
        // wrapped each of the 7 loads in its own s_and_saveexec / branch and zeroed the register first, and the row went
        // through 2 - 3 register copies per coordinate: ~110 instructions per step, now ~60.)
        double g2[2][NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) g2[0][r] = g2[1][r] = 0.0;
        const int last_j = (l + 16 * (NR - 1) < k) ? l + 16 * (NR - 1) : k - 1;
        auto load_row = [&](double (&dst)[NR], const double* __restrict__ p, const double* __restrict__ pl) {
#pragma unroll
            for (int r = 0; r < NR - 1; ++r) dst[r] = p[16 * r];
            dst[NR - 1] = pl[0];
        };
        while (true) {
            const bool go = cvalid && it < 100 && (tol / kd) > 1e-8;
            if (__ballot(go) == 0ull) break;
            ++ran;
            if (go) tol = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) cur[w] = 0u;
            // running row pointers (their addresses cannot be hoisted out of the sweep loop: written as Gc[(i + 1) k + ...]
            // hipcc kept the addresses of all 16 NR rows in registers, 344 VGPRs at NR = 7)
            const double* __restrict__ gp = Gc + l;          // row i of the running coordinate, pieces 0 .. NR - 2
            const double* __restrict__ gpl = Gc + last_j;    // ... its last piece
            double dn0 = dgc[0], dn1 = dgc[1];        // (G_ii, 1 / G_ii) of the coordinate to come
            bool have = go && (act[0] & 1u) != 0u;    // row 0 fetched ahead? (per column)
            if (have) load_row(g2[0], gp, gpl);
            static_for<16 * NR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int ir = i >> 4, il = i & 15;
                constexpr int pb = i & 1;
                bool run_i = i < k;
                if (i <= 16 * (NR - 1)) { asm volatile("" : "+s"(one)); run_i = one != 0; }   // opaque, always true (see nnls_lane.h)
                if (run_i) {
                    const bool have_i = have;
                    have = false;
                    if constexpr (i + 1 < 16 * NR) {   // row i + 1, in flight during this coordinate -- if its coordinate moved last sweep
                        const bool more = (i + 1 <= 16 * (NR - 1)) || (i + 1 < k);
                        constexpr int i1 = i + 1;
                        have = more && go && (act[i1 >> 5] & (1u << (i1 & 31))) != 0u;
                        if (have) load_row(g2[pb ^ 1], gp + k, gpl + k);
                    }
                    // fence: one row ahead, no more
                    __builtin_amdgcn_sched_barrier(0);
                    double bi, xi;
                    nnls_row_bcast2<il>(b[ir], x[ir], bi, xi);
                    const double gii = dn0, rii = dn1;
                    if (i + 1 < 16 * NR) { dn0 = dgc[2 * (i + 1)]; dn1 = dgc[2 * (i + 1) + 1]; }
                    const double diff0 = sgl_nnls_quotient(bi, gii, rii, any_irr);   // b_i / g_ii (Markstein, see nnls_lane.h)
                    double dpen;
                    const double nd = sgl_nnls_nd_strict(diff0, xi, go, L1, L2, dpen);
                    const bool moved = nd != 0.0;   // (x and tol change only with nd != 0; a stopped column has nd = 0)
                    if (__ballot(moved) != 0ull) {   // the coordinate moves in one of the four columns
                        double xn = xi;
                        sgl_nnls_apply(dpen, nd, xn, tol);
                        x[ir] = (l == il) ? xn : x[ir];
                        if (moved && !have_i) load_row(g2[pb], gp, gpl);
#pragma unroll
                        for (int r = 0; r < NR; ++r) b[r] = fma(g2[pb][r], nd, b[r]);
                        cur[i >> 5] |= moved ? (1u << (i & 31)) : 0u;
                    }
                    gp += k;
                    gpl += k;
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
#pragma unroll
            for (int w = 0; w < NW; ++w) act[w] = cur[w];
            it += go ? 1 : 0;
        }
        if (cvalid) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int j = l + 16 * r;
                if (j < k) X[col * k + j] = x[r];
            }
            if (l == 0) total_sweeps += it;
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

template <int NR>
static int launch_nnls_quad_global(hipStream_t s, const double* G, int64_t gstride, const double* B, double* X, const int64_t* col_nnz,
                                   int k, int64_t ncols, double L1, double L2, unsigned long long* sweep_counter) {
    const int64_t nquads = (ncols + 3) / 4;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(nquads, 256 * 4 * 8));
    nnls_quad_global_kernel<NR><<<dim3((unsigned)blocks), dim3(64), 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

