// nnls (src/singlet.cpp:229-250): cyclic coordinate-descent NNLS, one solve
// per column, <= 100 sweeps, stop when tol / k <= 1e-8.  FP64 VALU bound.
//
// Two mappings:
//  * nnls_lane_kernel<KP> (nnls_lane.h): ONE LANE PER COLUMN.  b[KP] and x[KP] live in VGPRs,
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
#include <atomic>
#include "nnls_static_for.h"
#include "nnls_quad_global.h"
#include <algorithm>
#include <cstdlib>
#include <utility>
#include <type_traits>

// ---- host side of the lane-per-column solve (kernels in nnls_lane.h) ----------------------
int k_nnls_lane_launch1(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                        int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g,
                        dim3 b);
int k_nnls_lane_launch2(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                        int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g,
                        dim3 b);
int k_nnls_lane_launch3(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                        int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g,
                        dim3 b);

int k_nnls_lane_launch4(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                        int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g,
                        dim3 b);

int k_nnls_half_launch(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                       int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps);
// the two-lanes-per-column solve as generated asm (kernels_nnls_half_asm.hip): padded rank it serves k with (0: none)
int nnls_half_asm_kp(int k, double L1);
int k_nnls_half_launch_asm(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                           int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps);
// the sweep as generated asm (kernels_nnls_asm.hip): the ranks it has instances for, same protocol as the lane kernels
bool nnls_lane_asm_has(int KP, double L1, int64_t ncols);
int k_nnls_lane_launch_asm(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                           int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g, dim3 b);

int64_t nnls_repack_min_cols() {
    // read on every call (cheap): tests lower it to drive small problems through the multi-pass path
    const char* e = getenv("SGL_NNLS_REPACK_MIN_COLS");
    const long long x = e ? atoll(e) : 0;
    return (int64_t)(x > 0 ? x : (1 << 18));
}

// Row stride of the padded Gram the lane kernel reads (nnls_lane.h): KP for the scalar-operand instances
// (KP <= 40, kernels_nnls_lane1.hip), KP rounded up to 16 for the vector-load + DPP instances (lane2).
int nnls_gram_stride(int KP) { return KP > 40 ? (KP + 15) / 16 * 16 : KP; }
// padded rank of the lane kernel instance serving rank k (0: no instance, use the wave kernel)
int nnls_lane_kp(int k) { return k <= 64 ? (k + 1) / 2 * 2 : (k <= SGL_LANE_NNLS_MAX_K ? (k + 7) / 8 * 8 : 0); }

// 64 < k <= 128: two lanes per column, everything in registers (nnls_half.h; above 104 x sits in the AGPR half of the
// register file); SGL_NNLS_NO_HALF=1 keeps the x-scratch instances
static bool nnls_use_half(int KP) { return KP > 64 && KP <= 128 && !getenv("SGL_NNLS_NO_HALF"); }
static bool nnls_needs_xt(int KP) { return KP > 64 && !nnls_use_half(KP); }

int nnls_scratch_alloc(NnlsScratch& sc, int64_t cap, int k_for_xt) {
    nnls_scratch_free(sc);
    if (cap <= 0) return SGL_OK;
    bool ok = true;
    // x scratch of the k > 64 instances; lists / per-column state only where re-packing is used
    if (nnls_needs_xt(nnls_lane_kp(k_for_xt))) ok = sgl_pool_malloc(&sc.xt, sizeof(double) * (size_t)cap * k_for_xt) == hipSuccess;
    const bool two_lane = k_for_xt > 64 && nnls_use_half(nnls_lane_kp(k_for_xt));   // re-packs from half the columns (k_nnls_lane)
    if (ok && cap >= ((two_lane && !getenv("SGL_NNLS_REPACK_MIN_COLS")) ? nnls_repack_min_cols() / 2 : nnls_repack_min_cols()))
        ok = sgl_pool_malloc(&sc.list[0], sizeof(int32_t) * cap) == hipSuccess && sgl_pool_malloc(&sc.list[1], sizeof(int32_t) * cap) == hipSuccess &&
             sgl_pool_malloc(&sc.counts, sizeof(uint32_t) * (SGL_NNLS_MAX_PASSES + 1)) == hipSuccess &&
             sgl_pool_malloc(&sc.it_state, (size_t)cap) == hipSuccess && sgl_pool_malloc(&sc.tol_state, sizeof(double) * cap) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        nnls_scratch_free(sc);
        sgl_set_error("NNLS scratch: out of device memory");
        return SGL_ENOMEM;
    }
    sc.cap = cap;
    return SGL_OK;
}

void nnls_scratch_free(NnlsScratch& sc) {
    if (sc.list[0]) (void)sgl_pool_free(sc.list[0]);
    if (sc.list[1]) (void)sgl_pool_free(sc.list[1]);
    if (sc.counts) (void)sgl_pool_free(sc.counts);
    if (sc.it_state) (void)sgl_pool_free(sc.it_state);
    if (sc.tol_state) (void)sgl_pool_free(sc.tol_state);
    if (sc.xt) (void)sgl_pool_free(sc.xt);
    if (sc.prev_it) (void)sgl_pool_free(sc.prev_it);
    if (sc.packed) (void)sgl_pool_free(sc.packed);
    if (sc.sort_ws) (void)sgl_pool_free(sc.sort_ws);
    sc = NnlsScratch();
}

// ---- packing by sweep count -------------------------------------------------------------------------------------
// Lanes of a wave run in lock-step, so a wave executes its slowest column's sweeps: 35 at config 3 (with the re-packing
// passes) where the columns need 27 - 29.  How many sweeps a column needs carries over from one ALS iteration to the
// next (correlation 0.57 - 0.79 on the oracle's counts), so the first pass takes the columns in DESCENDING order of the
// sweeps their previous solve needed (nnls_lane_kernel writes them to prev_it): neighbours in that order share a wave.
// A column's own arithmetic does not depend on where it is packed: bit-identical.  Counting sort on the device, three
// small kernels per solve (keys 0 .. 100 in a byte; the order inside a key is whatever the atomics give).
#define SGL_PACK_BINS 128
#define SGL_PACK_CPB 2048   // columns per block
__global__ __launch_bounds__(256) void nnls_pack_hist_kernel(const uint8_t* __restrict__ key, int64_t n, int nblocks, uint32_t* __restrict__ hist) {
    __shared__ unsigned h[SGL_PACK_BINS];
    if (threadIdx.x < SGL_PACK_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t c0 = (int64_t)blockIdx.x * SGL_PACK_CPB;
    for (int q = threadIdx.x; q < SGL_PACK_CPB; q += 256) {
        const int64_t c = c0 + q;
        if (c < n) atomicAdd(&h[key[c] & (SGL_PACK_BINS - 1)], 1u);
    }
    __syncthreads();
    // bin-major, longest first: entry (bin b, block) at [(BINS - 1 - b) * nblocks + block]
    if (threadIdx.x < SGL_PACK_BINS) hist[(size_t)(SGL_PACK_BINS - 1 - threadIdx.x) * nblocks + blockIdx.x] = h[threadIdx.x];
}
// exclusive scan of the BINS * nblocks counts in place (one workgroup; <= 128 * 512 entries at a million columns)
__global__ __launch_bounds__(1024) void nnls_pack_scan_kernel(uint32_t* __restrict__ hist, int total, uint32_t* __restrict__ n_out) {
    __shared__ unsigned part[1024];
    const int per = (total + 1023) / 1024;
    const int a = threadIdx.x * per, b = (a + per < total) ? a + per : total;
    unsigned s = 0;
    {   // (eight independent loads per step: one at a time the thread's ~60 strided reads were 90 us of pure latency per solve)
        int e = a;
        for (; e + 8 <= b; e += 8) {
            unsigned v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = hist[e + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; e < b; ++e) s += hist[e];
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned v = (threadIdx.x >= (unsigned)off) ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned run = part[threadIdx.x] - s;
    {
        int e = a;
        for (; e + 8 <= b; e += 8) {
            unsigned v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = hist[e + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) { hist[e + q] = run; run += v[q]; }
        }
        for (; e < b; ++e) { const unsigned c = hist[e]; hist[e] = run; run += c; }
    }
    if (threadIdx.x == 1023) *n_out = part[1023];
}
__global__ __launch_bounds__(256) void nnls_pack_scatter_kernel(const uint8_t* __restrict__ key, int64_t n, int nblocks,
                                                                const uint32_t* __restrict__ offs, int32_t* __restrict__ packed) {
    __shared__ unsigned cur[SGL_PACK_BINS];
    if (threadIdx.x < SGL_PACK_BINS) cur[threadIdx.x] = offs[(size_t)(SGL_PACK_BINS - 1 - threadIdx.x) * nblocks + blockIdx.x];
    __syncthreads();
    const int64_t c0 = (int64_t)blockIdx.x * SGL_PACK_CPB;
    for (int q = threadIdx.x; q < SGL_PACK_CPB; q += 256) {
        const int64_t c = c0 + q;
        if (c < n) packed[atomicAdd(&cur[key[c] & (SGL_PACK_BINS - 1)], 1u)] = (int32_t)c;
    }
}

int nnls_pack_alloc(NnlsScratch& sc, int64_t ncols) {
    if (sc.prev_it && sc.pack_cap >= ncols) return SGL_OK;
    if (sc.prev_it) (void)sgl_pool_free(sc.prev_it);
    if (sc.packed) (void)sgl_pool_free(sc.packed);
    if (sc.sort_ws) (void)sgl_pool_free(sc.sort_ws);
    sc.prev_it = nullptr; sc.packed = nullptr; sc.sort_ws = nullptr; sc.pack_cap = 0;
    const int64_t nblocks = (ncols + SGL_PACK_CPB - 1) / SGL_PACK_CPB;
    if (sgl_pool_malloc(&sc.prev_it, (size_t)ncols) != hipSuccess || sgl_pool_malloc(&sc.packed, sizeof(int32_t) * (size_t)ncols) != hipSuccess ||
        sgl_pool_malloc(&sc.sort_ws, sizeof(uint32_t) * ((size_t)SGL_PACK_BINS * nblocks + 4)) != hipSuccess ||
        hipMemset(sc.prev_it, 0, (size_t)ncols) != hipSuccess) {
        (void)hipGetLastError();
        sgl_set_error("NNLS packing scratch: out of device memory");
        return SGL_ENOMEM;
    }
    sc.pack_cap = ncols;
    return SGL_OK;
}

int k_nnls_lane(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsScratch* scr, bool pack_by_sweeps) {
    if (ncols <= 0) return SGL_OK;
    auto launch_lane = nnls_lane_asm_has(KP, L1, ncols) ? k_nnls_lane_launch_asm
                       : (KP <= 40) ? k_nnls_lane_launch1 : (KP <= 64 ? k_nnls_lane_launch2 : (KP <= 104 ? k_nnls_lane_launch3 : k_nnls_lane_launch4));
    const bool half = nnls_use_half(KP);
    const bool half_asm = half && nnls_half_asm_kp(k, L1) == KP;
    auto launch = [&](hipStream_t s_, const double* Gp_, int KP_, double* B_, double* X_, const int64_t* nz_, int k_, int64_t nc_, double L1_,
                      double L2_, unsigned long long* sw_, const NnlsPass& ps_, dim3 g_, dim3 b_) {
        return half_asm ? k_nnls_half_launch_asm(s_, Gp_, KP_, B_, X_, nz_, k_, nc_, L1_, L2_, sw_, ps_)
               : half ? k_nnls_half_launch(s_, Gp_, KP_, B_, X_, nz_, k_, nc_, L1_, L2_, sw_, ps_)
                    : launch_lane(s_, Gp_, KP_, B_, X_, nz_, k_, nc_, L1_, L2_, sw_, ps_, g_, b_);
    };
    if (nnls_needs_xt(KP) && (scr == nullptr || scr->xt == nullptr || scr->cap < ncols)) {
        sgl_set_error("k_nnls_lane: k > 64 needs the x scratch");
        return SGL_ESTATE;
    }
    double* xt = nnls_needs_xt(KP) ? scr->xt : nullptr;
    const dim3 g((unsigned)((ncols + 255) / 256)), b(256);
    // (the two-lane solves put 32 columns in a wave: the chip is as full at half the columns, and re-packing pays from there --
    //  nnls_h per 200 000 columns k = 100 6.56 -> 6.16 ms, k = 128 11.44 -> 10.27; scripts/r5/r5_step26.sh)
    const int64_t repack_min = (half && !getenv("SGL_NNLS_REPACK_MIN_COLS")) ? nnls_repack_min_cols() / 2 : nnls_repack_min_cols();
    const bool repack = scr != nullptr && scr->list[0] != nullptr && scr->cap >= ncols && ncols >= repack_min;
    // first pass in descending order of the previous solve's sweep counts (lane instances up to k = 64 and the generated two-lane
    // solve above; SGL_NNLS_NO_PACK: A/B)
    const bool pack = pack_by_sweeps && (!half || half_asm) && scr != nullptr && scr->prev_it != nullptr && scr->pack_cap >= ncols && ncols >= 65536 &&
                      !getenv("SGL_NNLS_NO_PACK");
    const int32_t* list0 = nullptr;
    const uint32_t* count0 = nullptr;
    if (pack) {
        const int nblocks = (int)((ncols + SGL_PACK_CPB - 1) / SGL_PACK_CPB);
        uint32_t* n_dev = scr->sort_ws + (size_t)SGL_PACK_BINS * nblocks;
        nnls_pack_hist_kernel<<<dim3((unsigned)nblocks), dim3(256), 0, s>>>(scr->prev_it, ncols, nblocks, scr->sort_ws);
        nnls_pack_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>(scr->sort_ws, SGL_PACK_BINS * nblocks, n_dev);
        nnls_pack_scatter_kernel<<<dim3((unsigned)nblocks), dim3(256), 0, s>>>(scr->prev_it, ncols, nblocks, scr->sort_ws, scr->packed);
        HIPCHK(hipGetLastError());
        list0 = scr->packed;
        count0 = n_dev;   // = ncols
    }
    uint8_t* prev_it = pack_by_sweeps && (!half || half_asm) && scr != nullptr && scr->prev_it != nullptr && scr->pack_cap >= ncols ? scr->prev_it : nullptr;
    if (!repack) {
        // fresh = 2: a packed single pass of one to two workgroups per CU may pair the longest waves with the shortest (kernels_nnls_asm.hip)
        const NnlsPass one = {list0, count0, nullptr, nullptr, nullptr, nullptr, 0, xt, ncols, (list0 != nullptr && !getenv("SGL_NNLS_NO_SNAKE")) ? 2 : 1, prev_it};
        SGLCHK(launch(s, Gpad, KP, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, one, g, b));
        HIPCHK(hipGetLastError());
        return SGL_OK;
    }
    // Passes over ever shorter lists; how many columns survive a pass is only known on the device, so
    // every pass is launched for the worst case and workgroups beyond the list return at once.  The
    // last pass, and any pass over a list too short to fill the GPU, runs its columns to the end.
    HIPCHK(hipMemsetAsync(scr->counts, 0, sizeof(uint32_t) * (SGL_NNLS_MAX_PASSES + 1), s));
    for (int p = 0; p < SGL_NNLS_MAX_PASSES; ++p) {
        const bool last = (p == SGL_NNLS_MAX_PASSES - 1);
        NnlsPass ps;
        ps.list = p ? scr->list[(p - 1) & 1] : list0;
        ps.count = p ? scr->counts + p : count0;
        ps.fresh = p ? 0 : 1;
        ps.prev_it = prev_it;
        ps.next_list = last ? nullptr : scr->list[p & 1];
        ps.next_count = last ? nullptr : scr->counts + p + 1;
        ps.it_state = scr->it_state;
        ps.tol_state = scr->tol_state;
        ps.final_below = (int32_t)std::min<int64_t>(64 * 1024, repack_min / 4);
        ps.xt = xt;
        ps.xt_stride = ncols;
        SGLCHK(launch(s, Gpad, KP, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps, g, b));
        HIPCHK(hipGetLastError());
    }
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
        double b[R], x[R], gd[R], rg[R];
        bool irr = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = lane + 64 * r;
            b[r] = (j < k) ? B[col * k + j] : 0.0;
            x[r] = (j < k) ? X[col * k + j] : 0.0;
            gd[r] = (j < k) ? Gc[(int64_t)j * k + j] : 1.0;
            rg[r] = 1.0 / gd[r];   // correctly rounded reciprocal of this column's diagonal, once per column
            irr = irr || !__builtin_isnormal(rg[r]);
        }
        const bool any_irr = __ballot(irr) != 0ull;   // a zero (or otherwise irregular) diagonal entry: the reference's own division there
        double tol = 1.0;
        int it = 0;
        for (; it < 100 && (tol / kd) > 1e-8; ++it) {
            tol = 0.0;
            for (int i = 0; i < k; ++i) {
                const int ir = i >> 6, il = i & 63;
                double bsel = b[0], xsel = x[0], gsel = gd[0], rsel = rg[0];
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    if (ir == r) { bsel = b[r]; xsel = x[r]; gsel = gd[r]; rsel = rg[r]; }
                }
                const double bi = rl64(bsel, il), xi = rl64(xsel, il), gii = rl64(gsel, il), rii = rl64(rsel, il);
                // b_i / g_ii from the reciprocal (Markstein correction; see nnls_lane.h), then l.235-247
                const double diff0 = sgl_nnls_quotient(bi, gii, rii, any_irr);
                double xn = xi, dpen;
                const double nd = sgl_nnls_nd_strict(diff0, xi, true, L1, L2, dpen);
                if (nd != 0.0) {  // wave-uniform; at rest: x, tol and b stay as they are
                    sgl_nnls_apply(dpen, nd, xn, tol);
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

template <int J>
__device__ __forceinline__ void nnls_row_bcast3(double bsrc, double xsrc, double rsrc, double& bi, double& xi, double& ri) {
    const int blo = __double2loint(bsrc), bhi = __double2hiint(bsrc), xlo = __double2loint(xsrc), xhi = __double2hiint(xsrc);
    const int rlo = __double2loint(rsrc), rhi = __double2hiint(rsrc);
    int r0, r1, r2, r3, r4, r5;
    asm("s_nop 1\n\t"
        "v_mov_b32_dpp %0, %6 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %7 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %2, %8 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %3, %9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %4, %10 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %5, %11 row_newbcast:%12 row_mask:0xf bank_mask:0xf"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5)
        : "v"(blo), "v"(bhi), "v"(xlo), "v"(xhi), "v"(rlo), "v"(rhi), "n"(J));
    bi = __hiloint2double(r1, r0);
    xi = __hiloint2double(r3, r2);
    ri = __hiloint2double(r5, r4);
}

// ---------------------------------------------------------------------------
// Per-column Gram (masked path), k <= 64: FOUR columns per wave, one 16-lane DPP row each ("quad").  In
// nnls_wave_kernel the ~60 instructions of a coordinate step that are uniform over the wave (two divisions,
// clamp logic, tol) are issued once per column; here one issue serves four columns, and a lane's share of the
// Gram row update is ceil(k / 16) FMAs instead of one.  The column's Gram is staged ONCE as a packed lower
// triangle in LDS (k (k + 1) / 2 doubles: 10 KB at k = 50, four per wave), so global memory sees it once
// instead of once per sweep; b_i, x_i and 1 / g_ii reach the row's lanes by DPP row_newbcast (nnls_row_bcast3), g_ii and
// the Gram row come from LDS.  Arithmetic and order are those of nnls_lane.h (the folded form of
// src/singlet.cpp:229-250), results bit-identical.
template <int NR>
__global__ __launch_bounds__(64) void nnls_quad_kernel(const double* __restrict__ G, int64_t gstride,
                                                       const double* __restrict__ B, double* __restrict__ X,
                                                       const int64_t* __restrict__ col_nnz, int k, int64_t ncols,
                                                       double L1, double L2, unsigned long long* __restrict__ sweep_counter) {
    extern __shared__ __attribute__((aligned(16))) char quad_lds_raw[];
    const int lane = threadIdx.x, grp = lane >> 4, l = lane & 15;
    const int TRI = k * (k + 1) / 2, TS = (TRI + 1) & ~1;
    double* tri = reinterpret_cast<double*>(quad_lds_raw) + grp * TS;   // tri[i (i + 1) / 2 + j] = a[i, j], j <= i
    const double kd = (double)k;
    int trij[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { const int j = l + 16 * r; trij[r] = (j < k) ? j * (j + 1) / 2 : 0; }
    long long total_sweeps = 0, ran_total = 0;
    const int64_t nquads = (ncols + 3) >> 2;
    for (int64_t quad = blockIdx.x; quad < nquads; quad += gridDim.x) {
        const int64_t col = quad * 4 + grp;
        const bool cvalid = col < ncols && (col_nnz == nullptr || col_nnz[col] != 0);
        const double* Gc = G + (cvalid ? col : 0) * gstride;
        __builtin_amdgcn_wave_barrier();
        if (cvalid) {   // the 16 lanes of the row stage their column's lower triangle
            for (int i = 0; i < k; ++i) {
                const int ti = i * (i + 1) / 2;
                for (int j = l; j <= i; j += 16) tri[ti + j] = Gc[(int64_t)i * k + j];
            }
        }
        double b[NR], x[NR], rg[NR];
        bool irr = false;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = l + 16 * r;
            const bool v = cvalid && j < k;
            b[r] = v ? B[col * k + j] : 0.0;
            x[r] = v ? X[col * k + j] : 0.0;
            rg[r] = v ? 1.0 / Gc[(int64_t)j * k + j] : 1.0;   // correctly rounded reciprocal of the diagonal
            irr = irr || !__builtin_isnormal(rg[r]);
        }
        const bool any_irr = __ballot(irr) != 0ull;   // a zero (or otherwise irregular) diagonal entry in one of the four columns
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        double tol = 1.0;
        int it = 0, ran = 0, one = 1;
        while (true) {
            const bool go = cvalid && it < 100 && (tol / kd) > 1e-8;
            if (__ballot(go) == 0ull) break;
            ++ran;
            if (go) tol = 0.0;
            static_for<16 * NR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int ir = i >> 4, il = i & 15;
                // an instance serves 16 (NR - 1) < k <= 16 NR: below that the test is always true; those coordinates
                // branch on an opaque always-true scalar instead (see nnls_lane.h)
                bool run_i = i < k;
                if (i <= 16 * (NR - 1)) { asm volatile("" : "+s"(one)); run_i = one != 0; }
                if (run_i) {
                    constexpr int ti = i * (i + 1) / 2;
                    double bi, xi, rii;
                    nnls_row_bcast3<il>(b[ir], x[ir], rg[ir], bi, xi, rii);
                    const double gii = tri[ti + i];
                    double g[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        if (r < ir) g[r] = tri[ti + l + 16 * r];                 // j < i for the whole row of lanes
                        else if (r > ir) g[r] = tri[trij[r] + i];                // j > i: a[j, i]
                        else g[r] = tri[(l <= il) ? (ti + l + 16 * r) : (trij[r] + i)];
                    }
                    const double diff0 = sgl_nnls_quotient(bi, gii, rii, any_irr);   // b_i / g_ii (Markstein, see nnls_lane.h)
                    double dpen;
                    const double nd = sgl_nnls_nd_strict(diff0, xi, go, L1, L2, dpen);
                    if (__ballot(nd != 0.0) != 0ull) {   // at rest in all four columns: x, tol and b stay as they are
                        double xn = xi;
                        sgl_nnls_apply(dpen, nd, xn, tol);
                        x[ir] = (l == il) ? xn : x[ir];
#pragma unroll
                        for (int r = 0; r < NR; ++r) b[r] = fma(g[r], nd, b[r]);
                    }
                }
            });
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

// ---------------------------------------------------------------------------
// SHARED Gram, FEW columns (round 5): four columns per wave against the one Gram of predict (src/singlet.cpp:333-347).
// The lane-per-column kernel is built for throughput -- 64 columns per wave, ~4150 instructions per sweep at k = 50 -- and a
// solve of a few thousand columns (a rank's gene block on an 8-GPU team: 3750 genes; the cells of pbmc3k: 2700) runs it on a
// few dozen waves, each alone on its SIMD: the solve then lasts as long as ONE wave's sweeps, 24 x 11 us.  Here a column has
// the 16 lanes of a DPP row (lane l holds coordinates l, l + 16, ...): a sweep is ~40 instructions per coordinate for four
// columns instead of ~83 for 64, i.e. SIXTEEN times the waves at less than half the sweep length, and a wave waits for the
// slowest of 4 columns instead of 64.  The Gram is staged once per workgroup as a full square (row stride 16 NR: row i of this
// lane's coordinates at immediate offsets, no triangle indexing -- nnls_quad_kernel with a zero column stride lost to that),
// (G_ii, 1 / G_ii) as pairs read by one uniform 16-byte read.  Arithmetic, order and stop test are the lane kernel's (b_i / G_ii
// by the Markstein form from the correctly rounded reciprocal, sgl_nnls_nd / sgl_nnls_apply, fma(G_ji, nd, b_j)): the same bits.
// Pays while the lane kernel would be latency-bound: up to ~8000 columns (sgl_nnls_shared).
template <int NR>
__global__ __launch_bounds__(256) void nnls_quad_shared_kernel(const double* __restrict__ G, const double* __restrict__ B, double* __restrict__ X,
                                                               const int64_t* __restrict__ col_nnz, int k, int64_t ncols, double L1, double L2,
                                                               unsigned long long* __restrict__ sweep_counter) {
    constexpr int GS = 16 * NR;
    extern __shared__ __attribute__((aligned(16))) char qs_lds_raw[];
    double* Gl = reinterpret_cast<double*>(qs_lds_raw);   // Gl[i * GS + j] = G[j, i] (= G[i, j]), 0 beyond the rank
    double* Dl = Gl + (size_t)k * GS;                      // (G_jj, 1 / G_jj)
    for (int e = threadIdx.x; e < k * GS; e += blockDim.x) {
        const int i = e / GS, j = e - i * GS;
        Gl[e] = (j < k) ? G[j + (size_t)k * i] : 0.0;
    }
    for (int j = threadIdx.x; j < GS; j += blockDim.x) {
        const double g = (j < k) ? G[j + (size_t)k * j] : 1.0;
        Dl[2 * j] = g;
        Dl[2 * j + 1] = 1.0 / g;   // correctly rounded reciprocal of the diagonal
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, grp = lane >> 4, l = lane & 15;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const double kd = (double)k;
    const double* __restrict__ gl = Gl + l;
    long long total_sweeps = 0, ran_total = 0;
    const int64_t nquads = (ncols + 3) >> 2;
    for (int64_t quad = wave; quad < nquads; quad += nwaves) {
        const int64_t col = quad * 4 + grp;
        const bool cvalid = col < ncols && (col_nnz == nullptr || col_nnz[col] != 0);
        double b[NR], x[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = l + 16 * r;
            const bool v = cvalid && j < k;
            b[r] = v ? B[col * k + j] : 0.0;
            x[r] = v ? X[col * k + j] : 0.0;
        }
        double tol = 1.0;
        int it = 0, ran = 0, one = 1;
        while (true) {
            const bool go = cvalid && it < 100 && (tol / kd) > 1e-8;
            if (__ballot(go) == 0ull) break;
            ++ran;
            if (go) tol = 0.0;
            double dn0 = Dl[0], dn1 = Dl[1];
            static_for<16 * NR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int ir = i >> 4, il = i & 15;
                bool run_i = i < k;
                if (i <= 16 * (NR - 1)) { asm volatile("" : "+s"(one)); run_i = one != 0; }   // opaque, always true (see nnls_lane.h)
                if (run_i) {
                    double g[NR];
#pragma unroll
                    for (int r = 0; r < NR; ++r) g[r] = gl[i * GS + 16 * r];
                    double bi, xi;
                    nnls_row_bcast2<il>(b[ir], x[ir], bi, xi);
                    const double gii = dn0, rii = dn1;
                    if (i + 1 < 16 * NR) { dn0 = Dl[2 * (i + 1)]; dn1 = Dl[2 * (i + 1) + 1]; }
                    const double q0 = bi * rii;
                    const double diff0 = fma(fma(-q0, gii, bi), rii, q0);   // b_i / G_ii (Markstein, see nnls_lane.h)
                    double dpen;
                    const double nd = sgl_nnls_nd(diff0, xi, go, L1, L2, dpen);
                    if (__ballot(nd != 0.0) != 0ull) {   // at rest in all four columns: x, tol and b stay as they are
                        double xn = xi;
                        sgl_nnls_apply(dpen, nd, xn, tol);
                        x[ir] = (l == il) ? xn : x[ir];
#pragma unroll
                        for (int r = 0; r < NR; ++r) b[r] = fma(g[r], nd, b[r]);
                    }
                }
            });
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
static int launch_nnls_quad_shared(hipStream_t s, const double* G, const double* B, double* X, const int64_t* col_nnz, int k, int64_t ncols,
                                   double L1, double L2, unsigned long long* sweep_counter) {
    const size_t lds = sizeof(double) * ((size_t)k * 16 * NR + 2 * 16 * NR);
    const int64_t nquads = (ncols + 3) / 4;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((nquads + 3) / 4, 256 * 8));
    nnls_quad_shared_kernel<NR><<<dim3((unsigned)blocks), dim3(256), lds, s>>>(G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// G: the k x k Gram (column-major, + 1e-15 on the diagonal), shared by all columns; B is read only.  k <= 64.
int k_nnls_quad_shared(hipStream_t s, const double* G, const double* B, double* X, const int64_t* col_nnz, int k, int64_t ncols,
                       double L1, double L2, unsigned long long* sweep_counter) {
    if (ncols <= 0) return SGL_OK;
    switch ((k + 15) / 16) {
        case 1: return launch_nnls_quad_shared<1>(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 2: return launch_nnls_quad_shared<2>(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 3: return launch_nnls_quad_shared<3>(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 4: return launch_nnls_quad_shared<4>(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        default: sgl_set_error("k_nnls_quad_shared: k=%d above 64", k); return SGL_EINVAL;
    }
}

template <int NR>
static int launch_nnls_quad(hipStream_t s, const double* G, int64_t gstride, const double* B, double* X, const int64_t* col_nnz,
                            int k, int64_t ncols, double L1, double L2, unsigned long long* sweep_counter) {
    const int TRI = k * (k + 1) / 2, TS = (TRI + 1) & ~1;
    const size_t lds = sizeof(double) * 4 * (size_t)TS;
    static std::atomic<bool> attr_set[64];   // per (function, device); several host threads may drive devices at once
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (lds > 48 * 1024 && (dev < 0 || dev >= 64 || !attr_set[dev])) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nnls_quad_kernel<NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(16, (160 * 1024) / (int64_t)lds));
    const int64_t nquads = (ncols + 3) / 4;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(nquads, 256 * per_cu * 4));
    nnls_quad_kernel<NR><<<dim3((unsigned)blocks), dim3(64), lds, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

int k_nnls_quad_global_big1(hipStream_t s, const double* G, int64_t gstride, const double* B, double* X, const int64_t* col_nnz, int k,
                            int64_t ncols, double L1, double L2, unsigned long long* sweep_counter);   // kernels_nnls_quad_big1.hip: k <= 192
int k_nnls_quad_global_big2(hipStream_t s, const double* G, int64_t gstride, const double* B, double* X, const int64_t* col_nnz, int k,
                            int64_t ncols, double L1, double L2, unsigned long long* sweep_counter);   // kernels_nnls_quad_big2.hip: k <= 256
static int k_nnls_quad_global_big(hipStream_t s, const double* G, int64_t gstride, const double* B, double* X, const int64_t* col_nnz, int k,
                                  int64_t ncols, double L1, double L2, unsigned long long* sweep_counter) {
    return k <= 192 ? k_nnls_quad_global_big1(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter)
                    : k_nnls_quad_global_big2(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
}

int k_nnls_quarter_part1(hipStream_t s, const double* G, const double* B, double* X, const int64_t* col_nnz, int k, int64_t ncols, double L1, double L2,
                         unsigned long long* sweep_counter, const int32_t* order, uint8_t* prev_it);   // kernels_nnls_quarter1.hip: k <= 192
int k_nnls_quarter_part2(hipStream_t s, const double* G, const double* B, double* X, const int64_t* col_nnz, int k, int64_t ncols, double L1, double L2,
                         unsigned long long* sweep_counter, const int32_t* order, uint8_t* prev_it);   // kernels_nnls_quarter2.hip: k <= 256
static int k_nnls_quarter(hipStream_t s, const double* G, const double* B, double* X, const int64_t* col_nnz, int k, int64_t ncols, double L1, double L2,
                          unsigned long long* sweep_counter, const int32_t* order = nullptr, uint8_t* prev_it = nullptr) {
    if (ncols <= 0) return SGL_OK;
    return k <= 192 ? k_nnls_quarter_part1(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, order, prev_it)
                    : k_nnls_quarter_part2(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, order, prev_it);
}

// the four-lane solve of a plain fit's H side (ranks 129 - 256, the whole shard's columns) with its 16-column waves packed by the
// sweep counts of the previous iteration's solve (the lane kernels' packing: counting sort on the device, then one launch)
int k_nnls_quarter_packed(hipStream_t s, const double* G, const double* B, double* X, const int64_t* col_nnz, int k, int64_t ncols, double L1,
                          double L2, unsigned long long* sweep_counter, const NnlsScratch* scr) {
    const bool pack = scr != nullptr && scr->prev_it != nullptr && scr->pack_cap >= ncols && !getenv("SGL_NNLS_NO_PACK");
    const int32_t* order = nullptr;
    if (pack) {
        const int nblocks = (int)((ncols + SGL_PACK_CPB - 1) / SGL_PACK_CPB);
        uint32_t* n_dev = scr->sort_ws + (size_t)SGL_PACK_BINS * nblocks;
        nnls_pack_hist_kernel<<<dim3((unsigned)nblocks), dim3(256), 0, s>>>(scr->prev_it, ncols, nblocks, scr->sort_ws);
        nnls_pack_scan_kernel<<<dim3(1), dim3(1024), 0, s>>>(scr->sort_ws, SGL_PACK_BINS * nblocks, n_dev);
        nnls_pack_scatter_kernel<<<dim3((unsigned)nblocks), dim3(256), 0, s>>>(scr->prev_it, ncols, nblocks, scr->sort_ws, scr->packed);
        HIPCHK(hipGetLastError());
        order = scr->packed;
    }
    return k_nnls_quarter(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, order, pack ? scr->prev_it : nullptr);
}

int k_nnls_percol(hipStream_t s, const double* G, int64_t gstride, const double* B, double* X, const int64_t* col_nnz,
                int k, int64_t ncols, double L1, double L2, unsigned long long* sweep_counter) {
    if (ncols <= 0) return SGL_OK;
    if (k > SGL_MAX_K) { sgl_set_error("k_nnls_percol: k=%d > %d", k, SGL_MAX_K); return SGL_EINVAL; }
    // per-column Grams: four columns per wave while four waves' triangles fit a CU's LDS (k <= 50).  Measured at
    // 30 000 x 100 000 (nnls_h, ms): k = 10: 1.1 vs 4.4 for nnls_wave_kernel, k = 30: 5.8 vs 14.0, k = 50: 21.5 vs
    // 27.1; at k = 64 (two waves per CU) 48.4 vs 38.5, so larger ranks stay on the wave kernel.  (env: A/B tests)
    // Round 4: once the global-memory solve below had lost its copies and per-load branches it overtakes this one where the
    // triangles leave one wave per SIMD: nnls_h per masked iteration at 30 000 x 200 000 (LDS -> global): k = 32: 6.5 -> 7.6,
    // 36: 8.6 -> 9.0, 40: 10.1 -> 9.7, 44: 12.7 -> 10.9, 48: 16.9 -> 11.6; on the 30 000 columns of the W side the LDS solve
    // stays ahead up to k = 44.  (SGL_NNLS_QUAD_GLOBAL_FROM: the first rank that takes the global solve -- A/B tests)
    // (Round 6: four LANES per column for per-column Grams too -- 16 columns per wave, each lane reading its KQ doubles of the
    //  column's Gram row -- was built, bit-identical, and measured SLOWER at every rank: nnls_h per masked iteration at 30 000 x
    //  200 000 k = 50 11.6 -> 18.5 ms, 100 36.3 -> 70.5 (profiles/r6_quarter_percol_ab.txt): these solves are bound by the reads of
    //  the Gram rows, which 16-lane rows fetch as 128-byte pieces at four waves per SIMD; taken out again.)
    const char* qmin = getenv("SGL_NNLS_QUAD_GLOBAL_MIN_COLS");   // (tests lower it to reach the long-launch choices with small problems)
    const int64_t long_launch = (qmin && atoll(qmin) > 0) ? atoll(qmin) : 65536;
    const char* qfrom = getenv("SGL_NNLS_QUAD_GLOBAL_FROM");
    const int global_from = (qfrom && atoi(qfrom) > 0) ? atoi(qfrom) : (ncols >= long_launch ? 40 : 47);
    if (gstride != 0 && k < global_from && 4 * 4 * sizeof(double) * (size_t)(((k * (k + 1) / 2) + 1) & ~1) <= 160 * 1024 && !getenv("SGL_NNLS_NO_QUAD")) {
        switch ((k + 15) / 16) {
            case 1: return launch_nnls_quad<1>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            case 2: return launch_nnls_quad<2>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            case 3: return launch_nnls_quad<3>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            default: return launch_nnls_quad<4>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        }
    }
    // per-column Grams above that, up to k = 112: four columns per wave on the Gram in global memory.  Measured per masked
    // iteration at 30 000 x 200 000 (nnls_h, ms; wave kernel -> this one): k = 56: 102 -> 50, 64: 119 -> 68, 80: 234 -> 122,
    // 100: 330 -> 245, 112: 370 -> 311; at k = 128 (eight row registers per lane) 416 -> 428: the wave kernel stays there.
    // (env: A/B tests)
    // Late round 3: those numbers were taken with 256 MB chunks of Grams (3 355 columns per launch at k = 100: a quarter of
    // the wave slots).  With chunks that fill the chip (sgl_mask_workspace) k = 113 ... 128 gains too when the launch has
    // columns enough (nnls_h k = 120: 198 -> 134 ms, 128: 216 -> 146; the 30 000 genes of the W side: 19.5 -> 24.0, so
    // short launches keep the wave kernel there).  SGL_NNLS_QUAD_GLOBAL_112=1: the old limit for every launch.
    const int quad_global_max_k = (ncols >= long_launch && !getenv("SGL_NNLS_QUAD_GLOBAL_112")) ? 128 : 112;
    // (instance NR serves 16 (NR - 1) < k <= 16 NR: it runs coordinates 0 .. 16 (NR - 1) unconditionally and prefetches their rows)
    if (gstride != 0 && k <= quad_global_max_k && !getenv("SGL_NNLS_NO_QUAD_GLOBAL")) {
        switch ((k + 15) / 16) {
            case 1: return launch_nnls_quad_global<1>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            case 2: return launch_nnls_quad_global<2>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            case 3: return launch_nnls_quad_global<3>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            case 4: return launch_nnls_quad_global<4>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            case 5: return launch_nnls_quad_global<5>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            case 6: return launch_nnls_quad_global<6>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            case 7: return launch_nnls_quad_global<7>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            default: return launch_nnls_quad_global<8>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        }
    }
    // ranks 129 - 256 against a SHARED Gram: four LANES per column, 16 columns per wave (nnls_quarter.h) from 8192 columns
    // on (SGL_NNLS_QUARTER_MIN_COLS; SGL_NNLS_NO_QUARTER=1: never -- A/B, bit-identity tests); shorter launches and per-column
    // Grams take the four-columns-per-wave solve below
    {
        const char* qm = getenv("SGL_NNLS_QUARTER_MIN_COLS");
        const int64_t quarter_min = (qm && atoll(qm) > 0) ? atoll(qm) : 8192;
        if (gstride == 0 && k > 128 && k <= 256 && ncols >= quarter_min && !getenv("SGL_NNLS_NO_QUARTER"))
            return k_nnls_quarter(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
    }
    // ranks 129 - 256, shared Gram or per-column Grams: four columns per wave, instances NR = 9 .. 16 (kernels_nnls_quad_big1 / 2.hip;
    // SGL_NNLS_NO_QUAD_BIG=1: the wave kernel below -- A/B, bit-identity tests)
    // (measured at 30 000 x 200 000, profiles/r6_k_above_128.txt: nnls_h k = 130 78 -> 41 ms, 200 140 -> 75, 256 192 -> 110; the 30 000
    //  columns of the W side gain up to k = 200 -- 8.8 -> 6.9 ms at k = 130 -- and lose above, where the instances spill: 20.1 -> 23.3
    //  at k = 256: short launches keep the wave kernel above k = 208)
    if (k > 128 && k <= 256 && (ncols >= long_launch || k <= 208) && !getenv("SGL_NNLS_NO_QUAD_GLOBAL") && !getenv("SGL_NNLS_NO_QUAD_BIG"))
        return k_nnls_quad_global_big(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
    int64_t blocks = (ncols + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 g((unsigned)blocks), b(256);
    const int R = (k + 63) / 64;
    switch (R) {
        case 1: nnls_wave_kernel<1><<<g, b, 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter); break;
        case 2: nnls_wave_kernel<2><<<g, b, 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter); break;
        case 3: nnls_wave_kernel<3><<<g, b, 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter); break;
        case 4: nnls_wave_kernel<4><<<g, b, 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter); break;
        default:   // ranks above 256 (the reference has no limit; SGL_MAX_K = 1024)
            if (R <= 8) nnls_wave_kernel<8><<<g, b, 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            else nnls_wave_kernel<16><<<g, b, 0, s>>>(G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
            break;
    }
    HIPCHK(hipGetLastError());
    return SGL_OK;
}
