// LDS-tiled sparse accumulate:  B[:, c] = sum_{(r, v) in column c} v * F[:, r]
// (predict, src/singlet.cpp:341-343) with the gathered operand F served from
// LDS instead of L2.
//
// Why: every non-zero needs k doubles of F (400 B at k = 50) against 12 B of
// matrix data, so the accumulate is bound by operand delivery, not by the HBM
// stream.  L2 delivers ~34 TB/s chip-wide, LDS ~150 TB/s.
//
// Layout ("re-blocked stream", built once per (matrix, k) by sgl_tiled_build):
//   * rows are cut into tiles of TR rows, TR * k * 8 B <= 128 KiB (one LDS tile);
//   * columns are cut into wave blocks of CW = 64 columns; a workgroup of 8 waves
//     owns 8 * CW columns and keeps their k-vectors in VGPRs (lane = factor
//     row, one f64 accumulator per column slot) across all row tiles;
//   * the non-zeros of (wave block wb, tile t) form one chunk, stored slot by
//     slot, each slot's run padded to a multiple of 4 entries and the chunk to
//     a multiple of 64 (x = 0 pads), as two coalesced arrays: roff (byte offset
//     of the row inside the LDS tile) and x.  Chunks are ordered (wb, t), so
//     one wave reads ONE linear stream;
//   * cnt[(wb * T + t) * CW + s] = number of 4-entry groups of slot s (the
//     chunk padding is booked on the last slot).
// Order of summation inside a column is the stored (ascending row) order, as in
// the reference; pads add x = 0 times a finite F entry.
//
// When the column count is too small to fill the chip (W-update: 30 k genes)
// the tile range is split over blockIdx.y; each split writes a partial k x ncol
// slab and acc_tiled_reduce sums the slabs in a fixed order.
#include "sgl_internal.h"
#include <utility>
#include <type_traits>

#define TILED_NW 8           // waves per workgroup (512 threads -> 256 VGPRs per lane)
#define TILED_CW 64          // column slots (FP64 accumulators) per wave
#define TILED_LDS_BYTES (128 * 1024)

template <typename F, int... Is>
__device__ __forceinline__ void t_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void t_static_for(F&& f) {
    t_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ double t_readlane_f64(double v, int lane) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffull), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---------------------------------------------------------------- build -----
// groups per (wb, t, slot) and entries per chunk
__global__ void tiled_count_kernel(const int64_t* __restrict__ seg, int64_t ncol, int T, int CW, int64_t nwb,
                                   uint8_t* __restrict__ cnt, int64_t* __restrict__ chunk_entries) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // u = wb * T + t
    if (u >= nwb * T) return;
    const int64_t wb = u / T;
    const int t = (int)(u - wb * T);
    int64_t tot = 0;
    for (int s = 0; s < CW; ++s) {
        const int64_t col = wb * CW + s;
        int g = 0;
        if (col < ncol) {
            const int64_t c = seg[(int64_t)(t + 1) * ncol + col] - seg[(int64_t)t * ncol + col];
            g = (int)((c + 3) >> 2);
        }
        if (s == CW - 1) g += (int)((16 - ((tot + g) & 15)) & 15);  // chunk = whole 64-entry steps
        cnt[u * CW + s] = (uint8_t)g;
        tot += g;
    }
    chunk_entries[u] = tot * 4;
}

// one wave per chunk (wb, t): copy / pad the CW segments
__global__ __launch_bounds__(256) void tiled_fill_kernel(const double* __restrict__ x, const int32_t* __restrict__ idx,
                                                         const int64_t* __restrict__ seg, int64_t ncol, int T, int CW,
                                                         int64_t nwb, int TR, int k, const uint8_t* __restrict__ cnt,
                                                         const int64_t* __restrict__ cstart,
                                                         uint32_t* __restrict__ sroff, double* __restrict__ sx) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = gw; u < nwb * T; u += nw) {
        const int64_t wb = u / T;
        const int t = (int)(u - wb * T);
        int64_t dst = cstart[u];
        for (int s = 0; s < CW; ++s) {
            const int n4 = 4 * (int)cnt[u * CW + s];
            if (n4 == 0) continue;
            const int64_t col = wb * CW + s;
            const int64_t a = seg[(int64_t)t * ncol + col], b = seg[(int64_t)(t + 1) * ncol + col];
            for (int q = lane; q < n4; q += 64) {
                uint32_t ro = 0;
                double xv = 0.0;
                if (a + q < b) {
                    ro = (uint32_t)(idx[a + q] - t * TR) * (uint32_t)(k * 8);
                    xv = x[a + q];
                }
                sroff[dst + q] = ro;
                sx[dst + q] = xv;
            }
            dst += n4;
        }
    }
}

template <typename T_>
static int t_alloc(T_** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)p, count * sizeof(T_));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        sgl_set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T_), hipGetErrorString(e));
        return SGL_ENOMEM;
    }
    return SGL_OK;
}

void sgl_tiled_free(DevTiled& S) {
    if (S.roff) (void)hipFree(S.roff);
    if (S.x) (void)hipFree(S.x);
    if (S.cstart) (void)hipFree(S.cstart);
    if (S.cnt) (void)hipFree(S.cnt);
    if (S.part) (void)hipFree(S.part);
    S = DevTiled();
}

int sgl_tiled_build(sgl_ctx* c, const DevCSC& M, int k, DevTiled& S) {
    sgl_tiled_free(S);
    hipStream_t s = c->stream;
    S.k = k;
    S.CW = TILED_CW;
    int TR = TILED_LDS_BYTES / (k * 8);
    TR = TR / 8 * 8;
    if (TR > 952) TR = 952;  // groups per slot (+ <= 15 chunk-padding groups) must fit a byte
    if (TR < 8) { sgl_set_error("tiled accumulate: k=%d too large", k); return SGL_EINVAL; }
    S.TR = TR;
    S.T = (int)((M.nrow + TR - 1) / TR);
    S.nwb = ((int64_t)M.ncol + S.CW - 1) / S.CW;
    S.ncol = M.ncol;
    S.nrow = M.nrow;
    const int64_t nchunks = S.nwb * S.T;

    // segment starts per (tile, column)
    DevCSC tmp = M;
    tmp.tile_rows = TR;
    tmp.ntiles = S.T;
    tmp.seg = nullptr;
    SGLCHK(t_alloc(&tmp.seg, (size_t)(S.T + 1) * (size_t)M.ncol));
    int rc = k_build_segments(s, tmp);
    int64_t* chunk_entries = nullptr;
    if (rc == SGL_OK) rc = t_alloc(&chunk_entries, (size_t)nchunks);
    if (rc == SGL_OK) rc = t_alloc(&S.cnt, (size_t)nchunks * S.CW);
    if (rc == SGL_OK) rc = t_alloc(&S.cstart, (size_t)nchunks + 1);
    if (rc == SGL_OK) {
        tiled_count_kernel<<<dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s>>>(tmp.seg, M.ncol, S.T, S.CW, S.nwb,
                                                                                         S.cnt, chunk_entries);
        if (hipGetLastError() != hipSuccess) rc = SGL_EHIP;
    }
    if (rc == SGL_OK) rc = k_exclusive_scan(c, chunk_entries, S.cstart, nchunks);
    if (rc == SGL_OK) rc = k_scan_total(s, chunk_entries, S.cstart, nchunks);
    int64_t E = 0;
    if (rc == SGL_OK) {
        if (hipMemcpyAsync(&E, S.cstart + nchunks, sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) rc = SGL_EHIP;
    }
    S.E = E;
    // + 256 entries of slack: the kernel prefetches a 64-entry batch past the end
    if (rc == SGL_OK) rc = t_alloc(&S.roff, (size_t)E + 256);
    if (rc == SGL_OK) rc = t_alloc(&S.x, (size_t)E + 256);
    if (rc == SGL_OK) {
        if (hipMemsetAsync(S.roff + E, 0, 256 * sizeof(uint32_t), s) != hipSuccess ||
            hipMemsetAsync(S.x + E, 0, 256 * sizeof(double), s) != hipSuccess) rc = SGL_EHIP;
    }
    if (rc == SGL_OK && nchunks > 0) {
        int64_t blocks = (nchunks + 3) / 4;
        if (blocks > 256 * 64) blocks = 256 * 64;
        tiled_fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(M.x, M.i, tmp.seg, M.ncol, S.T, S.CW, S.nwb, TR, k,
                                                                       S.cnt, S.cstart, S.roff, S.x);
        if (hipGetLastError() != hipSuccess) rc = SGL_EHIP;
    }
    // split of the tile range over blockIdx.y so that the grid fills 256 CUs (1 workgroup per CU)
    const int64_t nwg_x = (S.nwb + TILED_NW - 1) / TILED_NW;
    int R = 1;
    if (nwg_x < 1024) {
        double best = -1.0;
        const int rmin = (int)std::max<int64_t>(1, (512 + nwg_x - 1) / nwg_x);
        for (int r = rmin; r <= std::min<int64_t>(S.T, 4 * rmin); ++r) {
            const int tpr = (S.T + r - 1) / r;
            const int reff = (S.T + tpr - 1) / tpr;  // ranges actually non-empty
            const double wgs = (double)nwg_x * reff;
            const double eff = wgs / (ceil(wgs / 256.0) * 256.0);
            if (eff > best + 1e-9) { best = eff; R = reff; }
        }
    }
    S.tiles_per_range = (S.T + R - 1) / R;
    S.R = (S.T + S.tiles_per_range - 1) / S.tiles_per_range;
    if (rc == SGL_OK && S.R > 1) rc = t_alloc(&S.part, (size_t)S.R * (size_t)k * (size_t)M.ncol);
    hipError_t e = hipStreamSynchronize(s);
    if (tmp.seg) (void)hipFree(tmp.seg);
    if (chunk_entries) (void)hipFree(chunk_entries);
    if (rc == SGL_OK && e != hipSuccess) { sgl_set_error("tiled build failed: %s", hipGetErrorString(e)); rc = SGL_EHIP; }
    if (rc != SGL_OK) sgl_tiled_free(S);
    return rc;
}

// ---------------------------------------------------------------- kernel ----
// Broadcasting the wave-uniform (roff, x) of each non-zero is the VALU cost
// that decides this kernel: three v_readlane per non-zero cost ~30 cycles
// (measured on MI355X: VALU-bound at 35 cyc/nz/SIMD).  Instead every 16-lane
// ROW of the wave holds the same 16 entries (lane l loads entry l & 15), and
// DPP row_newbcast:j -- the only DPP mode gfx950 allows on FP64 ALU ops --
// feeds entry j to all lanes inside the consuming instruction itself:
//     v_add_u32_dpp  addr, roff, lane*8   row_newbcast:j     (LDS address)
//     ds_read_b64    w, addr                                (F[lane, row])
//     v_fmac_f64_dpp acc, x, w            row_newbcast:j     (acc += x_j * w)
// Two VALU instructions per non-zero; the kernel is then bound by the
// ds_read_b64 rate (2 cycles per wave instruction per CU).
//
// The wave walks its stream in 16-entry batches, four statically named batch
// register sets in flight (loaded by inline asm right after a set is consumed,
// waited with a counted vmcnt: loads return in order).  The column slot of a
// 4-entry group is dynamic, so the running accumulator `a` is swapped with the
// slot's home register (a uniform-indexed VGPR array, s_set_gpr_idx) only
// when the slot changes.  The loop body is a few hundred instructions.
typedef double d16 __attribute__((ext_vector_type(16)));

template <int J>
__device__ __forceinline__ int dpp_addr(uint32_t roff, int lane8) {
    int a;
    asm("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(roff), "v"(lane8), "n"(J));
    return a;
}
template <int J>
__device__ __forceinline__ void dpp_fmac(double& acc, double x, double w) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w), "n"(J));
}

// Accumulator home registers.  The 64 FP64 accumulators of a wave live in v[128:255], OUTSIDE the
// compiler's register allocation: the kernel is compiled with amdgpu_num_vgpr(128) (hipcc may only
// use v0..v127) and an asm clobber of v255 makes the kernel descriptor allocate all 256.  They are
// reached with VGPR index mode (s_set_gpr_idx_on: M0 = 2 * slot is added to the register number), the
// one form of dynamic register addressing the ISA has.  Letting hipcc index a register array
// dynamically was tried first: it either moved the array to scratch memory or copied whole
// 32-register vectors around every slot change.
__device__ __forceinline__ void acc_store(int slot2, double v) {
    asm volatile("s_set_gpr_idx_on %0, gpr_idx(DST)\n\tv_mov_b32 v128, %1\n\tv_mov_b32 v129, %2\n\ts_set_gpr_idx_off"
                 :: "s"(slot2), "v"(__double2loint(v)), "v"(__double2hiint(v)) : "memory");
}
__device__ __forceinline__ double acc_load(int slot2) {
    int lo, hi;
    asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC0)\n\tv_mov_b32 %0, v128\n\tv_mov_b32 %1, v129\n\ts_set_gpr_idx_off"
                 : "=v"(lo), "=v"(hi) : "s"(slot2) : "memory");
    return __hiloint2double(hi, lo);
}
template <int CW>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(128))) void acc_tiled_kernel(const uint32_t* __restrict__ sroff, const double* __restrict__ sx,
                                                        const int64_t* __restrict__ cstart,
                                                        const uint8_t* __restrict__ cnt, int T, int64_t nwb,
                                                        const double* __restrict__ F, int k, int TR, int64_t nrow,
                                                        int tiles_per_range, double* __restrict__ Bout, int64_t ncol) {
    static_assert(CW == 64, "accumulators are v[128:255]");
    asm volatile("" ::: "v255");  // make the kernel descriptor allocate 256 VGPRs
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    const int lane = threadIdx.x & 63;
    const int l16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wb = (int64_t)blockIdx.x * TILED_NW + wave;
    const int t0 = blockIdx.y * tiles_per_range;
    const int t1 = (t0 + tiles_per_range < T) ? (t0 + tiles_per_range) : T;
    const bool wact = wb < nwb;
    // LDS byte address of this lane's factor row inside the tile (raw 32-bit LDS addresses are used
    // below so that no per-entry base add is emitted)
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __attribute__((address_space(3))) const double lds_cdouble;
    const int lane8 = (int)(uint32_t)(uintptr_t)(lds_char*)smem + lane * 8;

    for (int q = 0; q < CW; ++q) acc_store(2 * q, 0.0);

    int64_t pos = 0;   // stream position (entries) of the next batch to LOAD
    int cnt_next = 0;
    if (wact) {
        pos = cstart[wb * T + t0];
        cnt_next = (int)cnt[(wb * T + t0) * CW + lane];
    }
    // four batch register sets; set q always holds batch (4*i + q) of the stream.  (Plain named
    // variables and macros, no lambdas: a by-reference capture of the accumulator vectors makes
    // them escape and land in scratch memory.)
    uint32_t er0 = 0, er1 = 0, er2 = 0, er3 = 0;
    double ex0 = 0.0, ex1 = 0.0, ex2 = 0.0, ex3 = 0.0;
#define TILED_ISSUE(ER, EX)                                                                       \
    do {                                                                                          \
        const uint32_t* pr_ = sroff + (pos + l16);                                                \
        const double* px_ = sx + (pos + l16);                                                     \
        asm volatile("global_load_dword %0, %1, off" : "=v"(ER) : "v"(pr_) : "memory");          \
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(EX) : "v"(px_) : "memory");        \
        pos += 16;                                                                                \
    } while (0)
    if (wact) {
        TILED_ISSUE(er0, ex0);
        TILED_ISSUE(er1, ex1);
        TILED_ISSUE(er2, ex2);
        TILED_ISSUE(er3, ex3);
    }

    for (int t = t0; t < t1; ++t) {
        const int cntv = cnt_next;  // lane s: 4-entry groups of slot s in this tile
        if (wact && t + 1 < t1) cnt_next = (int)cnt[(wb * T + t + 1) * CW + lane];
        int tot = cntv;             // total groups of the chunk (a multiple of 16)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
        const int nb = __builtin_amdgcn_readfirstlane(tot) >> 2;  // 16-entry batches, a multiple of 4

        // stage rows [t*TR, ...) of F (contiguous k*rows doubles) into LDS
        const int64_t row0 = (int64_t)t * TR;
        const int rows = (int)((nrow - row0 < TR) ? (nrow - row0) : TR);
        const int n = rows * k;
        const double* __restrict__ src = F + row0 * k;
        // Each thread moves up to NST 16-byte pieces of the tile, in two rounds of NST/2.  Round 1
        // loads are issued before the barrier (they overlap the tail of the previous tile's work of
        // other waves).  Reading one double past an odd-sized tile is harmless: factor buffers carry
        // 2 doubles of slack, the LDS tile 512 B.
        constexpr int NST = TILED_LDS_BYTES / 16 / (64 * TILED_NW);
        constexpr int HST = NST / 2;
        double2 stg[HST];
#pragma unroll
        for (int j = 0; j < HST; ++j) {
            const int e = ((int)threadIdx.x + j * 64 * TILED_NW) * 2;
            stg[j] = double2{0.0, 0.0};
            if (e < n) stg[j] = *reinterpret_cast<const double2*>(src + e);
        }
        __syncthreads();  // everyone is done reading the previous tile
#pragma unroll
        for (int j = 0; j < HST; ++j) {
            const int e = ((int)threadIdx.x + j * 64 * TILED_NW) * 2;
            if (e < n) *reinterpret_cast<double2*>(tile + e) = stg[j];
        }
#pragma unroll
        for (int j = 0; j < HST; ++j) {
            const int e = ((int)threadIdx.x + (j + HST) * 64 * TILED_NW) * 2;
            stg[j] = double2{0.0, 0.0};
            if (e < n) stg[j] = *reinterpret_cast<const double2*>(src + e);
        }
#pragma unroll
        for (int j = 0; j < HST; ++j) {
            const int e = ((int)threadIdx.x + (j + HST) * 64 * TILED_NW) * 2;
            if (e < n) *reinterpret_cast<double2*>(tile + e) = stg[j];
        }
        __syncthreads();
        if (!wact) continue;

        int s = -1, rem = 0;
        double a = 0.0;
        // One 16-entry batch: all 16 LDS addresses and ds_reads are issued first (16 reads in flight
        // per wave keep the LDS pipe busy; with only 2 waves per SIMD a 4-deep group at a time is
        // latency-bound: measured 8 cyc/nz/CU against the 2 cyc/nz/CU the LDS can do), the batch
        // registers are refilled for four batches ahead, then the 16 FMAs follow in stored order with
        // the (wave-uniform) column-slot bookkeeping in front of every 4-entry group.
#define TILED_AW(J) \
        const int a##J##_ = dpp_addr<J>(rr_, lane8); \
        const double w##J##_ = *(lds_cdouble*)(uintptr_t)(uint32_t)a##J##_;
#define TILED_SLOT()                                                                              \
    do {                                                                                          \
        if (rem == 0) { /* next column slot (uniform) */                                          \
            if (s >= 0) acc_store(2 * s, a);                                                      \
            do {                                                                                  \
                ++s;                                                                              \
                rem = __builtin_amdgcn_readlane(cntv, s);                                         \
            } while (rem == 0);                                                                   \
            a = acc_load(2 * s);                                                                  \
        }                                                                                         \
        --rem;                                                                                    \
    } while (0)
        // set q's two loads are complete once at most the six younger ones (the three other sets)
        // remain; s_nop: keep any VALU write of a batch register two wait states from the DPP reads
#define TILED_SET(ER, EX)                                                                         \
    do {                                                                                          \
        /* the batch registers pass THROUGH the wait, so no consumer can be scheduled above it */   \
        asm volatile("s_waitcnt vmcnt(6)\n\ts_nop 1" : "+v"(ER), "+v"(EX) : : "memory");          \
        const uint32_t rr_ = ER;                                                                  \
        const double xx_ = EX;                                                                    \
        TILED_AW(0) TILED_AW(1) TILED_AW(2) TILED_AW(3) TILED_AW(4) TILED_AW(5) TILED_AW(6) TILED_AW(7)       \
        TILED_AW(8) TILED_AW(9) TILED_AW(10) TILED_AW(11) TILED_AW(12) TILED_AW(13) TILED_AW(14) TILED_AW(15) \
        TILED_SLOT();                                                                             \
        dpp_fmac<0>(a, xx_, w0_); dpp_fmac<1>(a, xx_, w1_); dpp_fmac<2>(a, xx_, w2_); dpp_fmac<3>(a, xx_, w3_);         \
        TILED_SLOT();                                                                             \
        dpp_fmac<4>(a, xx_, w4_); dpp_fmac<5>(a, xx_, w5_); dpp_fmac<6>(a, xx_, w6_); dpp_fmac<7>(a, xx_, w7_);         \
        TILED_SLOT();                                                                             \
        dpp_fmac<8>(a, xx_, w8_); dpp_fmac<9>(a, xx_, w9_); dpp_fmac<10>(a, xx_, w10_); dpp_fmac<11>(a, xx_, w11_);     \
        TILED_SLOT();                                                                             \
        dpp_fmac<12>(a, xx_, w12_); dpp_fmac<13>(a, xx_, w13_); dpp_fmac<14>(a, xx_, w14_); dpp_fmac<15>(a, xx_, w15_); \
        TILED_ISSUE(ER, EX); /* refill this set with the batch four ahead */                      \
    } while (0)
        for (int b = 0; b < nb; b += 4) {
            TILED_SET(er0, ex0);
            TILED_SET(er1, ex1);
            TILED_SET(er2, ex2);
            TILED_SET(er3, ex3);
        }
        if (s >= 0) acc_store(2 * s, a);
    }
    if (wact) {
        double* out = Bout + (size_t)blockIdx.y * (size_t)k * (size_t)ncol;
        for (int q = 0; q < CW; ++q) {
            const double v = acc_load(2 * q);
            const int64_t col = wb * CW + q;
            if (col < ncol && lane < k) out[col * k + lane] = v;
        }
    }
}

__global__ void acc_tiled_reduce_kernel(const double* __restrict__ part, int R, int64_t n, double* __restrict__ B) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        double s = part[e];
        for (int r = 1; r < R; ++r) s += part[(size_t)r * n + e];
        B[e] = s;
    }
}

int k_acc_tiled(hipStream_t s, const DevTiled& S, const double* F, double* B) {
    if (S.ncol <= 0) return SGL_OK;
    const size_t lds = (size_t)S.TR * S.k * 8 + 512;
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&acc_tiled_kernel<TILED_CW>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, TILED_LDS_BYTES + 512));
        attr_set = true;
    }
    const int64_t nwg_x = (S.nwb + TILED_NW - 1) / TILED_NW;
    double* out = (S.R > 1) ? S.part : B;
    acc_tiled_kernel<TILED_CW><<<dim3((unsigned)nwg_x, (unsigned)S.R), dim3(64 * TILED_NW), lds, s>>>(
        S.roff, S.x, S.cstart, S.cnt, S.T, S.nwb, F, S.k, S.TR, S.nrow, S.tiles_per_range, out, S.ncol);
    HIPCHK(hipGetLastError());
    if (S.R > 1) {
        const int64_t n = (int64_t)S.k * S.ncol;
        int64_t blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        acc_tiled_reduce_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(S.part, S.R, n, B);
        HIPCHK(hipGetLastError());
    }
    return SGL_OK;
}
