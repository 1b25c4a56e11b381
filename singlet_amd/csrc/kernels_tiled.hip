// LDS-tiled sparse accumulate:  B[:, c] = sum_{(r, v) in column c} v * F[:, r]
// (predict, src/singlet.cpp:341-343) with the gathered operand F served from
// LDS instead of L2.
//
// Why: every non-zero needs k doubles of F (400 B at k = 50) against 12 B of
// matrix data, so the accumulate is bound by operand delivery, not by the HBM
// stream.  L2 delivers ~34 TB/s chip-wide, LDS ~150 TB/s.
// Measured on MI355X (scripts/ubench/lds_rate.hip): the LDS retires one wave
// instruction per ~1.7 ns per CU whether it is ds_read_b64 or ds_read_b128, so
// the kernel is built around ds_read_b128: a lane reads TWO consecutive factor
// rows of F, 32 lanes cover k <= 64, and the two half-waves work on two
// different non-zeros at once (two non-zeros per LDS instruction).
//
// Layout ("re-blocked stream", built once per (matrix, k) by sgl_tiled_build):
//   * rows are cut into tiles of TR rows, TR * KS * 8 B <= 160 KiB - 512 B (one LDS
//     tile; KS = k rounded up to even so every row starts 16-byte aligned);
//   * columns are cut into wave blocks of 64 columns; a workgroup of 8 waves
//     owns 512 columns and keeps their k-vectors in VGPRs across all row tiles.
//     Lanes 0-31 of a wave accumulate column p of the block, lanes 32-63 column
//     32 + p ("column pair" p = 0..31), lane l holding factors 2(l&31), +1;
//   * per (wave block wb, tile t) -- a chunk -- the entries of the pair's two
//     columns are stored as two half-streams that advance in lockstep: the runs
//     of both columns are padded (x = 0 entries) to the same multiple of 4, the
//     chunk to a multiple of 32 entries per half.  64 consecutive stream slots
//     hold 32 entries of the A half then 32 of the B half, so one coalesced
//     64-lane load fetches a "set".  Two arrays: roff (byte offset of the row
//     inside the LDS tile) and x.  Chunks are ordered (wb, t): one wave reads
//     ONE linear stream;
//   * cnt[(wb * T + t) * 32 + p] = number of 4-entry groups of pair p (chunk
//     padding is booked on the last pair).
// Inside a column the products are added in stored (ascending row) order, as
// in the reference; pads add x = 0 times a finite F entry.
//
// Ranks up to 32 use FOUR columns per LDS instruction instead of two (NSL = 4, round 4): a lane still holds two
// factors, so one 16-lane DPP row covers the rank and the four rows of a wave work on four columns (a column QUAD);
// a set of 64 stream slots holds 16 entries of each of the quad's columns -- the operand layout row_newbcast wants,
// so a set needs no preparation -- a wave owns 32 quads = 128 columns, and the LDS tile rows are 256 B apart whatever
// the rank (every tile row starts on bank 0: the 16-lane groups of ds_read_b128 mix two DPP rows and stay
// conflict-free only then).  Everything below is written for NSL column slots per instruction (2 or 4).
//
// When the column count is too small to fill the chip (W-update: 30 k genes)
// the tile range is split over blockIdx.y; each split writes a partial k x ncol
// slab and acc_tiled_reduce sums the slabs in a fixed order.
#include "sgl_internal.h"
#include <atomic>
#include <hipcub/hipcub.hpp>
#include <stdlib.h>
#include <utility>
#include <type_traits>

#define TILED_NW 8           // waves per workgroup (512 threads -> 256 VGPRs per lane)
#define TILED_CW 64          // columns per wave
#define TILED_NP 32          // column pairs per wave
#define TILED_LDS_BYTES (160 * 1024 - 512)  // the whole 160 KiB of a CU minus the read-past-the-row slack
#define TILED_SLACK 4096     // stream entries readable past the end (ring prefetch: 256 - 512 entries, + the L2 prefetch laps beyond it)

// ---------------------------------------------------------------- build -----
// groups per (wb, t, pair) and entries per chunk
// Slot (half h, pair p) of wave block wb holds the column at SORTED position wb * 64 + 2p + h: the columns are taken in
// the order of descending non-zero count (perm), so the two columns of a pair are neighbours in that order and the
// runs they are padded to are of similar length whatever the skew of the matrix.
__device__ __forceinline__ int64_t tiled_slot_col(const int32_t* __restrict__ perm, int64_t ncol, int64_t wb, int h, int p, int nsl) {
    const int64_t pos = wb * (TILED_NP * nsl) + nsl * p + h;
    if (pos >= ncol) return -1;
    return perm ? (int64_t)perm[pos] : pos;
}

__global__ void tiled_count_kernel(const int64_t* __restrict__ seg, const int32_t* __restrict__ perm, int64_t ncol, int T, int64_t nwb,
                                   uint8_t* __restrict__ cnt, int64_t* __restrict__ chunk_entries, int nsl) {
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // u = wb * T + t
    if (u >= nwb * T) return;
    const int64_t wb = u / T;
    const int t = (int)(u - wb * T);
    int64_t tot = 0;
    for (int p = 0; p < TILED_NP; ++p) {
        int64_t n = 0;
        for (int h = 0; h < nsl; ++h) {
            const int64_t col = tiled_slot_col(perm, ncol, wb, h, p, nsl);
            if (col >= 0) {
                const int64_t c = seg[(int64_t)(t + 1) * ncol + col] - seg[(int64_t)t * ncol + col];
                n = c > n ? c : n;
            }
        }
        int g = (int)((n + 3) >> 2);
        const int gps = 16 / nsl;   // groups (of 4 entries per slot) in a 64-slot set: whole sets per chunk
        if (p == TILED_NP - 1) g += (int)((gps - ((tot + g) & (gps - 1))) & (gps - 1));
        cnt[u * TILED_NP + p] = (uint8_t)g;
        tot += g;
    }
    chunk_entries[u] = tot * 4 * nsl;  // all slots
}

// one wave per chunk (wb, t): copy / pad the runs of the 32 column pairs
__global__ __launch_bounds__(256) void tiled_fill_kernel(const double* __restrict__ x, const int32_t* __restrict__ idx,
                                                         const int64_t* __restrict__ seg, const int32_t* __restrict__ perm,
                                                         int64_t ncol, int T, int64_t nwb, int TR, int row_bytes,
                                                         const uint8_t* __restrict__ cnt,
                                                         const int64_t* __restrict__ cstart,
                                                         uint32_t* __restrict__ sroff, double* __restrict__ sx,
                                                         int masked, uint64_t seed, SglDiv inv_density, int mask_t,
                                                         int64_t col_off, int64_t row_off, int nsl) {
    const int lane = threadIdx.x & 63;
    const int eps = 64 / nsl, eps_sh = nsl == 4 ? 4 : 5;   // entries of one column slot in a 64-slot set
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = gw; u < nwb * T; u += nw) {
        const int64_t wb = u / T;
        const int t = (int)(u - wb * T);
        const int64_t c0 = cstart[u];
        int P = 0;  // position in the half-streams
        for (int p = 0; p < TILED_NP; ++p) {
            const int n4 = 4 * (int)cnt[u * TILED_NP + p];
            if (n4 == 0) continue;
            for (int h = 0; h < nsl; ++h) {
                const int64_t col = tiled_slot_col(perm, ncol, wb, h, p, nsl);
                int64_t a = 0, b = 0;
                if (col >= 0) { a = seg[(int64_t)t * ncol + col]; b = seg[(int64_t)(t + 1) * ncol + col]; }
                for (int q = lane; q < n4; q += 64) {
                    uint32_t ro = 0;
                    double xv = 0.0;
                    if (a + q < b) {
                        const int32_t r = idx[a + q];
                        ro = (uint32_t)(r - t * TR) * (uint32_t)row_bytes;
                        xv = x[a + q];
                        if (masked) {  // predict_mask leaves the drawn entries out (src/singlet.cpp:449-457): x = 0 adds +0 * F
                            const uint64_t gc = (uint64_t)(col + col_off), gr = (uint64_t)(r + row_off);
                            if (mask_t ? sgl_draw(seed, gr, gc, inv_density) : sgl_draw(seed, gc, gr, inv_density)) xv = 0.0;
                        }
                    }
                    const int pos = P + q;
                    const int64_t dst = c0 + (int64_t)(pos >> eps_sh) * 64 + h * eps + (pos & (eps - 1));
                    if (sroff != nullptr) sroff[dst] = ro;
                    sx[dst] = xv;
                }
            }
            P += n4;
        }
    }
}

// schedule table of the chunk loop (gen_acc_tiled.py, GenTab): one u16 per group of four entry tuples, in stream order --
// the M0 word (destination-relative register index 4 * unit, 0x8000 = VDST_REL) of the column unit the group belongs
// to; the padding groups at the end of a chunk are booked on the last unit, like their entries.  One wave per chunk.
__global__ __launch_bounds__(256) void tiled_gtab_kernel(const uint8_t* __restrict__ cnt, const int64_t* __restrict__ cstart,
                                                         int64_t nchunks, int nsl, uint16_t* __restrict__ gtab) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t u = gw; u < nchunks; u += nw) {
        int64_t g0 = cstart[u] / (4 * nsl);
        for (int p = 0; p < TILED_NP; ++p) {
            const int n = (int)cnt[u * TILED_NP + p];
            for (int q = lane; q < n; q += 64) gtab[g0 + q] = (uint16_t)(0x8000u | (unsigned)(4 * p));
            g0 += n;
        }
    }
}

template <typename T_>
static int t_alloc(T_** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = sgl_pool_malloc((void**)p, count * sizeof(T_));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        sgl_set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T_), hipGetErrorString(e));
        return SGL_ENOMEM;
    }
    return SGL_OK;
}

// grow-only buffer: keeps the allocation when it is large enough (3 % headroom on growth: the stream size moves
// by a few per cent with the rank)
template <typename T_>
static int t_reserve(T_** p, size_t* cap, size_t count) {
    if (*p != nullptr && *cap >= count) return SGL_OK;
    if (*p) (void)sgl_pool_free(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = count + count / 32 + 64;
    SGLCHK(t_alloc(p, want));
    *cap = want;
    return SGL_OK;
}

void sgl_tiled_free(DevTiled& S) {
    if (S.seg) (void)sgl_pool_free(S.seg);
    if (S.perm) (void)sgl_pool_free(S.perm);
    if (S.roff) (void)sgl_pool_free(S.roff);
    if (S.x) (void)sgl_pool_free(S.x);
    if (S.cstart) (void)sgl_pool_free(S.cstart);
    if (S.cnt) (void)sgl_pool_free(S.cnt);
    if (S.gtab) (void)sgl_pool_free(S.gtab);
    if (S.part) (void)sgl_pool_free(S.part);
    if (S.xm) (void)sgl_pool_free(S.xm);
    S = DevTiled();
}

// SGL_TILED_SORT=0 keeps the columns in matrix order (the round-2 layout; kept for A/B measurements and the tests)
static bool tiled_sort_columns() {
    const char* e = getenv("SGL_TILED_SORT");
    return !(e && *e == '0');
}

__global__ void tiled_col_keys_kernel(const int64_t* __restrict__ p, int64_t ncol, uint32_t* __restrict__ keys, int32_t* __restrict__ iota) {
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncol; c += (int64_t)gridDim.x * blockDim.x) {
        keys[c] = (uint32_t)(p[c + 1] - p[c]);
        iota[c] = (int32_t)c;
    }
}

// S.perm[pos] = column at position pos of the descending-count order
static int tiled_build_perm(sgl_ctx* c, const DevCSC& M, DevTiled& S) {
    hipStream_t s = c->stream;
    const int64_t n = M.ncol;
    SGLCHK(t_reserve(&S.perm, &S.cap_perm, (size_t)n));
    uint32_t *keys = nullptr, *keys_out = nullptr;
    int32_t* iota = nullptr;
    void* tmp = nullptr;
    int rc = t_alloc(&keys, (size_t)n);
    if (rc == SGL_OK) rc = t_alloc(&keys_out, (size_t)n);
    if (rc == SGL_OK) rc = t_alloc(&iota, (size_t)n);
    if (rc == SGL_OK && n > 0) {
        const int64_t blocks = std::min<int64_t>((n + 255) / 256, 4096);
        tiled_col_keys_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(M.p, n, keys, iota);
        size_t tmp_bytes = 0;
        if (hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tmp_bytes, keys, keys_out, iota, S.perm, n, 0, 32, s) != hipSuccess) rc = SGL_EHIP;
        if (rc == SGL_OK && sgl_pool_malloc(&tmp, tmp_bytes ? tmp_bytes : 1) != hipSuccess) { (void)hipGetLastError(); rc = SGL_ENOMEM; }
        if (rc == SGL_OK && hipcub::DeviceRadixSort::SortPairsDescending(tmp, tmp_bytes, keys, keys_out, iota, S.perm, n, 0, 32, s) != hipSuccess) rc = SGL_EHIP;
    }
    // share of the non-zeros held by the heaviest workgroup's columns (the first 8 x 64 of the order): the tile-range
    // split below sizes its work units by it
    static_assert(TILED_CW == 64, "pair layout: 64 columns per wave");
    std::vector<uint32_t> top_v((size_t)TILED_NW * TILED_CW * 2);   // quad layout: 128 columns per wave
    uint32_t* top = top_v.data();
    const int64_t ntop = std::min<int64_t>(n, (int64_t)top_v.size());
    if (rc == SGL_OK && ntop > 0 && hipMemcpyAsync(top, keys_out, sizeof(uint32_t) * (size_t)ntop, hipMemcpyDeviceToHost, s) != hipSuccess) rc = SGL_EHIP;
    const hipError_t e = hipStreamSynchronize(s);
    if (rc == SGL_OK && e == hipSuccess) {
        int64_t tn = 0, tn2 = 0;
        for (int64_t q = 0; q < ntop; ++q) {
            if (q < TILED_NW * TILED_CW) tn += top[q];
            tn2 += top[q];
        }
        S.top_share = M.nnz > 0 ? (double)tn / (double)M.nnz : 0.0;
        S.top_share4 = M.nnz > 0 ? (double)tn2 / (double)M.nnz : 0.0;
    }
    if (keys) (void)sgl_pool_free(keys);
    if (keys_out) (void)sgl_pool_free(keys_out);
    if (iota) (void)sgl_pool_free(iota);
    if (tmp) (void)sgl_pool_free(tmp);
    if (rc == SGL_OK && e != hipSuccess) rc = SGL_EHIP;
    if (rc != SGL_OK) { sgl_set_error("tiled build: sorting the columns by their non-zero count failed"); return rc; }
    S.perm_nnz = M.nnz;
    return SGL_OK;
}

void sgl_trace_setup(const char* what);   // singlet_hip.hip (SGL_TRACE_SETUP=1)

int sgl_tiled_build(sgl_ctx* c, const DevCSC& M, int k, DevTiled& S) {
    hipStream_t s = c->stream;
    // ranks up to 32: four columns per LDS instruction (SGL_TILED_NO_QUAD=1: the pair layout at every rank; A/B, tests)
    const int nsl = (k <= 32 && !getenv("SGL_TILED_NO_QUAD")) ? 4 : 2;
    // same matrix (any change of it frees the streams), same part size: the stream is still valid
    if (S.built && S.k == k && S.NSL == nsl && S.ncol == M.ncol && S.nrow == M.nrow && S.src_nnz == M.nnz && (S.perm != nullptr) == tiled_sort_columns())
        return SGL_OK;
    S.built = false;
    S.xm_mask_t = -1;   // the masked value array follows the stream layout
    S.k = k;
    S.NSL = nsl;
    S.CW = TILED_NP * nsl;
    // LDS row stride in doubles.  Pair layout: k rounded up to even (rows start 16-byte aligned).  Quad layout: 32 for
    // every rank -- all tile rows must start on the same bank (header comment)
    const int KS = nsl == 4 ? 32 : ((k + 1) & ~1);
    S.KS = KS;
    int TR = TILED_LDS_BYTES / (KS * 8);
    TR = TR / 8 * 8;
    if (TR > 984) TR = 984;  // groups per pair (+ <= 7 chunk-padding groups) must fit a byte
    if (TR < 8 || k > 64) { sgl_set_error("tiled accumulate: k=%d unsupported", k); return SGL_EINVAL; }
    S.nwb = ((int64_t)M.ncol + S.CW - 1) / S.CW;
    // A small matrix cannot fill the chip with LDS-sized tiles even when every tile is a range of its own (pbmc3k: 3 column
    // groups x 22 tiles on the cell side, 14 x 5 on the gene side): shorter tiles -- at least 128 rows, so that a column still
    // brings a few entries per tile -- until column groups x tiles reach the 256 CUs.  (SGL_TILED_FULL_TILES=1: tests of the
    // chunk loop's extremes -- a column dense over a whole LDS-sized tile.)
    {
        const int64_t groups = (S.nwb + TILED_NW - 1) / TILED_NW, T0 = ((int64_t)M.nrow + TR - 1) / TR;
        if (groups * T0 < 256 && !getenv("SGL_TILED_FULL_TILES")) {
            const int64_t want = std::min<int64_t>((256 + groups - 1) / groups, ((int64_t)M.nrow + 127) / 128);
            if (want > T0) TR = std::min<int>(TR, (int)(((((int64_t)M.nrow + want - 1) / want) + 7) / 8 * 8));
        }
    }
    S.TR = TR;
    S.T = (int)((M.nrow + TR - 1) / TR);
    S.ncol = M.ncol;
    S.nrow = M.nrow;
    S.src_nnz = M.nnz;
    const int64_t nchunks = S.nwb * S.T;

    // column order of the stream: descending non-zero count (stable: ties in matrix order); depends on the matrix
    // only, so a rank sweep computes it once
    int rc = SGL_OK;
    if (!tiled_sort_columns()) {
        if (S.perm) { (void)sgl_pool_free(S.perm); S.perm = nullptr; S.cap_perm = 0; }
    } else if (!S.perm || S.perm_nnz != M.nnz || S.cap_perm < (size_t)M.ncol) {
        rc = tiled_build_perm(c, M, S);
    }
    sgl_trace_setup("  stream: column order");
    // segment starts per (tile, column); kept for the masked value array
    if (rc == SGL_OK) rc = t_reserve(&S.seg, &S.cap_seg, (size_t)(S.T + 1) * (size_t)M.ncol);
    sgl_trace_setup("  stream: segment buffer reserved");
    DevCSC tmp = M;
    tmp.tile_rows = TR;
    tmp.ntiles = S.T;
    tmp.seg = S.seg;
    if (rc == SGL_OK) rc = k_build_segments(s, tmp);
    int64_t* chunk_entries = nullptr;
    if (rc == SGL_OK) rc = t_alloc(&chunk_entries, (size_t)nchunks);
    if (rc == SGL_OK) rc = t_reserve(&S.cnt, &S.cap_cnt, (size_t)nchunks * TILED_NP);
    if (rc == SGL_OK) rc = t_reserve(&S.cstart, &S.cap_cstart, (size_t)nchunks + 1);
    if (rc == SGL_OK) {
        tiled_count_kernel<<<dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s>>>(S.seg, S.perm, M.ncol, S.T, S.nwb, S.cnt,
                                                                                         chunk_entries, nsl);
        if (hipGetLastError() != hipSuccess) { sgl_set_error("tiled build: count kernel launch failed"); rc = SGL_EHIP; }
    }
    if (rc == SGL_OK) rc = k_exclusive_scan(c, chunk_entries, S.cstart, nchunks);
    if (rc == SGL_OK) rc = k_scan_total(s, chunk_entries, S.cstart, nchunks);
    int64_t E = 0;
    if (rc == SGL_OK) {
        if (hipMemcpyAsync(&E, S.cstart + nchunks, sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { sgl_set_error("tiled build: reading the stream size failed"); rc = SGL_EHIP; }
    }
    S.E = E;
    sgl_trace_setup("  stream: segments, counts, scan");
    // + TILED_SLACK entries of slack: the kernel prefetches its ring (four or eight 64-entry sets) past the end
    if (rc == SGL_OK) rc = t_reserve(&S.roff, &S.cap_roff, (size_t)E + TILED_SLACK);
    if (rc == SGL_OK) rc = t_reserve(&S.x, &S.cap_x, (size_t)E + TILED_SLACK);
    sgl_trace_setup("  stream: roff / x buffers reserved");
    if (rc == SGL_OK) {
        if (hipMemsetAsync(S.roff + E, 0, TILED_SLACK * sizeof(uint32_t), s) != hipSuccess ||
            hipMemsetAsync(S.x + E, 0, TILED_SLACK * sizeof(double), s) != hipSuccess) { sgl_set_error("tiled build: clearing the stream slack failed"); rc = SGL_EHIP; }
    }
    if (rc == SGL_OK && nchunks > 0) {
        int64_t blocks = (nchunks + 3) / 4;
        if (blocks > 256 * 64) blocks = 256 * 64;
        tiled_fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(M.x, M.i, S.seg, S.perm, M.ncol, S.T, S.nwb, TR, KS * 8,
                                                                       S.cnt, S.cstart, S.roff, S.x, 0, 0, sgl_div_make(1), 0, 0, 0, nsl);
        if (hipGetLastError() != hipSuccess) { sgl_set_error("tiled build: fill kernel launch failed"); rc = SGL_EHIP; }
    }
    // the chunk loop's schedule table: E / (4 nsl) groups + slack (a lap of 32 groups is loaded one ahead)
    const size_t ngroups = (size_t)(E / (4 * nsl));
    if (rc == SGL_OK) rc = t_reserve(&S.gtab, &S.cap_gtab, ngroups + 256);
    if (rc == SGL_OK) {
        if (hipMemsetAsync(S.gtab + ngroups, 0, 256 * sizeof(uint16_t), s) != hipSuccess) { sgl_set_error("tiled build: clearing the schedule slack failed"); rc = SGL_EHIP; }
    }
    if (rc == SGL_OK && nchunks > 0) {
        int64_t blocks = (nchunks + 3) / 4;
        if (blocks > 256 * 64) blocks = 256 * 64;
        tiled_gtab_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(S.cnt, S.cstart, nchunks, nsl, S.gtab);
        if (hipGetLastError() != hipSuccess) { sgl_set_error("tiled build: schedule kernel launch failed"); rc = SGL_EHIP; }
    }
    // split of the tile range over blockIdx.y so that the grid fills 256 CUs (1 workgroup per CU)
    // Two reasons to split: too few column groups to fill the chip (W-update: 30 k genes = 59 workgroups), and -- with
    // the columns sorted by count -- a heaviest workgroup that alone would outlast the average CU's whole share (skewed
    // genes: the top 512 of 30 000 held 11 % of the non-zeros and the pass took 22 ms instead of 11): its tile range is
    // cut until one unit is about a third of a CU's share; workgroups are dispatched heaviest first.
    const int64_t nwg_x = (S.nwb + TILED_NW - 1) / TILED_NW;
    int R = 1;
    const int r_fill = nwg_x < 1024 ? (int)std::max<int64_t>(1, (512 + nwg_x - 1) / nwg_x) : 1;
    // (only when that workgroup really stands out: on i.i.d. columns its share is 1 / nwg_x and the fill rule decides,
    // with the column groups running fastest -- sizing by balance there cost 8 GB of factor-tile traffic per pass)
    const double top_share = nsl == 4 ? S.top_share4 : S.top_share;   // of the heaviest workgroup's 512 / 1024 columns
    const bool skewed = S.perm && top_share * (double)nwg_x > 1.5;
    const int r_bal = skewed ? (int)std::min<double>(256.0, ceil(top_share * 256.0 / 0.35)) : 1;
    if (r_bal > r_fill) {
        double best = -1.0;
        // (a fill factor beyond the tile count: every tile its own range -- round 3 left the range whole there, "such a matrix is
        // tiny", and the gene side of pbmc3k -- 14 column groups, 5 tiles, half of the non-zeros in the first group -- ran 1.07 ms)
        const int rmin = std::max(r_fill, (int)std::min<int64_t>(r_bal, std::max(1, S.T)));
        if (rmin > S.T) R = std::max(1, S.T);
        for (int r = rmin; r <= std::min<int64_t>(S.T, rmin + rmin / 4 + 4); ++r) {
            const int tpr = (S.T + r - 1) / r;
            const int reff = (S.T + tpr - 1) / tpr;  // ranges actually non-empty
            const double wgs = (double)nwg_x * reff;
            const double eff = wgs / (ceil(wgs / 256.0) * 256.0);
            if (eff > best + 1e-9) { best = eff; R = reff; }
        }
    } else if (r_fill > 1 && S.T > 1 && !getenv("SGL_TILED_OLD_SPLIT")) {
        // (matrices of every size: for most of round 4 the range stayed whole below 4 M entries -- "a pass takes microseconds
        // either way" -- and pbmc3k, 2.3 M non-zeros in 3 column groups of the quad layout, ran its 22 tiles on THREE CUs:
        // 0.83 / 1.07 ms per pass where config 2, with 22 times the non-zeros, takes 0.21.  A split adds the ranges' partial
        // sums: equal to the whole range up to rounding, and SGL_TILED_RANGES=1 keeps the reference's order for the tests that
        // compare bits.)
        // Too few column groups to fill the chip: cut the tile range into R pieces (sizes floor / ceil of T / R).  Round 4:
        // R by a cost model instead of "the count of workgroups nearest a multiple of 256" -- that rule took R = T at
        // BASELINE config 2 in the quad layout (49 column groups x 32 tiles: 1568 workgroups of ONE tile each, every
        // one clearing and staging 160 KB and writing a 245 KB slab: 0.41 ms per pass; 5 ranges: see profiles/).
        //   time(R) = rounds x (largest unit x t_tile + t_wg) + t_reduce(R),   rounds = ceil(groups x R / 256 CUs)
        // t_tile = one workgroup's entry tuples of one tile at the measured 2.4 ns per tuple per CU + 2.5 us of staging,
        // t_wg = 6 us per workgroup (launch, zeroing, output), t_reduce = the slabs written and read back at 4 TB/s.
        const double t_tile = (double)E / (double)nsl / ((double)nwg_x * (double)S.T) * 2.4e-9 + 2.5e-6;
        const double t_wg = 6e-6;
        double best = 1e300;
        for (int r = 1; r <= std::min<int64_t>(S.T, 64); ++r) {
            const double wgs = (double)nwg_x * r;
            const double rounds = ceil(wgs / 256.0);
            const int largest = (S.T + r - 1) / r;
            const double cost = rounds * ((double)largest * t_tile + t_wg) + (r > 1 ? (double)r * (double)k * (double)M.ncol * 16.0 / 4e12 : 0.0);
            if (cost < best * (1.0 - 1e-9)) { best = cost; R = r; }
        }
    } else if (r_fill > 1 && getenv("SGL_TILED_OLD_SPLIT")) {   // rounds 1 - 3: the workgroup count nearest a multiple of 256 (A/B)
        double best = -1.0;
        for (int r = r_fill; r <= std::min<int64_t>(S.T, 4 * r_fill); ++r) {
            const int tpr = (S.T + r - 1) / r;
            const int reff = (S.T + tpr - 1) / tpr;
            const double wgs = (double)nwg_x * reff;
            const double eff = wgs / (ceil(wgs / 256.0) * 256.0);
            if (eff > best + 1e-9) { best = eff; R = reff; }
        }
    }
    if (const char* fr = getenv("SGL_TILED_RANGES")) {   // tests: force the slab path (or the whole range) on small matrices
        if (atoi(fr) > 0) R = atoi(fr);
    }
    S.range_fastest = r_bal > r_fill;
    S.R = std::max(1, std::min(R, std::max(1, S.T)));   // range y = tiles [y T / R, (y + 1) T / R): none empty
    S.tiles_per_range = (S.T + S.R - 1) / S.R;
    // Tail split (round 4).  With enough column groups to fill the chip the range stays whole (R = 1), but the workgroups then
    // run in rounds of 256: config 3's H side has 1954 of them = 7.63 rounds -- the last round keeps 37 % of the CUs idle for a
    // whole workgroup's 74 tiles (4.6 % of the pass).  Only the workgroups of that last, partly filled round get their tile
    // range cut into tail_R pieces (dispatched last: blockIdx.x ascends), so that the pieces fill whole rounds of their own;
    // only their columns go through slabs.  Same cost model as above.
    S.tail_wg0 = -1;
    S.tail_R = 1;
    if (S.R == 1 && !S.range_fastest && nwg_x > 256 && S.T > 1 && E >= (4ll << 20) && !getenv("SGL_TILED_NO_TAIL")) {
        const int64_t tail = nwg_x % 256;
        if (tail > 0) {
            const double t_tile = (double)E / (double)nsl / ((double)nwg_x * (double)S.T) * 2.4e-9 + 2.5e-6;
            const double t_wg = 6e-6;
            const double tail_cols = (double)tail * TILED_NW * S.CW;
            double best = (double)S.T * t_tile + t_wg;   // left whole: one more round of full length
            int Rt = 1;
            for (int r = 2; r <= std::min<int64_t>(S.T, 16); ++r) {
                const double rounds = ceil((double)tail * r / 256.0);
                const double cost = rounds * ((double)((S.T + r - 1) / r) * t_tile + t_wg) + (double)r * (double)k * tail_cols * 16.0 / 4e12;
                if (cost < best * (1.0 - 1e-9)) { best = cost; Rt = r; }
            }
            if (Rt > 1) { S.tail_wg0 = nwg_x - tail; S.tail_R = Rt; }
        }
    }
    if (rc == SGL_OK && S.R > 1) rc = t_reserve(&S.part, &S.cap_part, (size_t)S.R * (size_t)k * (size_t)M.ncol);
    if (rc == SGL_OK && S.tail_R > 1)
        rc = t_reserve(&S.part, &S.cap_part, (size_t)S.tail_R * (size_t)k * (size_t)((nwg_x - S.tail_wg0) * TILED_NW * S.CW));
    hipError_t e = hipStreamSynchronize(s);
    if (chunk_entries) (void)sgl_pool_free(chunk_entries);
    if (rc == SGL_OK && e != hipSuccess) { sgl_set_error("tiled build failed: %s", hipGetErrorString(e)); rc = SGL_EHIP; }
    if (rc != SGL_OK) { sgl_tiled_free(S); return rc; }
    S.built = true;
    ++S.builds;
    return rc;
}

// ------------------------------------------------------- masked value array --
// The cross-validation mask of c_ard_nmf is fixed for a fit (seed, inv_density): predict_mask leaves the drawn
// entries out of the right-hand sides (src/singlet.cpp:449-457).  Instead of hashing every entry in every
// iteration (acc_kernel<MASK>), a second VALUE array of the entry stream is built once per fit with x = 0 at
// the drawn entries -- the tiled kernel then runs unchanged on it (a zero adds +0 * F: the sums are those of
// skipping the entry, as with the stream's pads).  8 B per stored entry on top of the 12.
// S.xm = the stream's value array for the mask (seed, inv_density); rebuilt only when the mask changes.  The hash
// runs inside the fill kernel (no temporary copy of the matrix values: at config-5 size that copy was a 12 GB
// allocation per fit and orientation).
int sgl_tiled_mask_values(sgl_ctx* c, const DevCSC& M, DevTiled& S, uint64_t seed, uint64_t inv_density, int mask_t,
                          int64_t col_off, int64_t row_off) {
    if (!S.built || !S.seg) { sgl_set_error("masked values: no entry stream"); return SGL_ESTATE; }
    if (S.xm && S.xm_seed == seed && S.xm_inv == inv_density && S.xm_mask_t == mask_t) return SGL_OK;
    hipStream_t s = c->stream;
    S.xm_mask_t = -1;
    SGLCHK(t_reserve(&S.xm, &S.cap_xm, (size_t)S.E + TILED_SLACK));
    HIPCHK(hipMemsetAsync(S.xm + S.E, 0, TILED_SLACK * sizeof(double), s));
    const int64_t nchunks = S.nwb * S.T;
    if (nchunks > 0) {
        const int KS = S.KS;
        int64_t blocks = (nchunks + 3) / 4;
        if (blocks > 256 * 64) blocks = 256 * 64;
        tiled_fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(M.x, M.i, S.seg, S.perm, M.ncol, S.T, S.nwb, S.TR, KS * 8, S.cnt,
                                                                       S.cstart, nullptr, S.xm, 1, seed, sgl_div_make(inv_density),
                                                                       mask_t, col_off, row_off, S.NSL);
        HIPCHK(hipGetLastError());
    }
    S.xm_seed = seed; S.xm_inv = inv_density; S.xm_mask_t = mask_t;
    return SGL_OK;
}

// ---------------------------------------------------------------- kernel ----
// Broadcasting the (roff, x) of each entry to the 32 lanes that work on it is
// the VALU cost that decides this kernel: three v_readlane per non-zero cost
// ~30 cycles (measured: VALU-bound at 35 cyc/nz/SIMD).  Instead the 16-lane
// ROWS of the wave hold 16 entries each -- rows 0,1 the same 16 entries of the A
// half, rows 2,3 sixteen of the B half -- and DPP row_newbcast:j (the only DPP
// mode gfx950 allows on FP64 ALU ops) feeds entry j to all lanes of its rows
// inside the consuming instruction itself:
//     v_add_u32_dpp  addr, roff, lane16        row_newbcast:j   (LDS address)
//     ds_read_b128   w, addr                                   (F[2l, 2l+1 ; row])
//     v_fmac_f64_dpp acc0, x, w.x              row_newbcast:j   (acc += x_j * w)
//     v_fmac_f64_dpp acc1, x, w.y              row_newbcast:j
// A 64-entry set (one coalesced load per array) becomes two such 16-pair
// batches with ONE v_permlane16_swap per dword: rows [A0 A1 B0 B1] ->
// [A0 A0 B0 B0] and [A1 A1 B1 B1].
//
// Register plan.  The 32 pairs x 2 FP64 accumulators of a wave live in v[128:255], OUTSIDE the compiler's register
// allocation, and are updated IN PLACE: VGPR index mode is on inside the chunk loop and M0 (destination-relative,
// 0x8000 | 4 * pair) selects the running pair's registers for the FMAs (gen_acc_tiled.py).  Letting hipcc index
// a register array dynamically was tried first: it either moved the array to scratch memory or copied whole
// 32-register vectors around every slot change.  (Rounds 1 - 2 ran the loop as compiler-scheduled C++ with the
// running pair swapped in and out of fixed registers: same arithmetic, 6 % slower; see DESIGN.md.)
//
// Four stream sets are in flight per wave (3 KiB), refilled right after a set is prepared and waited with a
// counted vmcnt (loads return in order); which ring slot is next is the wave-uniform `phase`.

// accumulators of pair p -> (v0, v1), for the output stage
__device__ __forceinline__ void acc_load(int idx4, double& v0, double& v1) {
    int a, b, c, d;
    asm volatile(
        "s_set_gpr_idx_on %4, gpr_idx(SRC0)\n\tv_mov_b32 %0, v128\n\tv_mov_b32 %1, v129\n\tv_mov_b32 %2, v130\n\t"
        "v_mov_b32 %3, v131\n\ts_set_gpr_idx_off"
        : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "s"(idx4) : "memory");
    v0 = __hiloint2double(b, a);
    v1 = __hiloint2double(d, c);
}

#include "acc_tiled_gen.inc"

// The chunk loop is the hand-scheduled inline asm of gen_acc_tiled.py (register plan there).  The compiler's budget
// is v0..v63 (amdgpu_waves_per_eu(8, 8) caps its allocation at 512 / 8 registers); the clobber makes the kernel
// descriptor allocate all 256: v64..v255 belong to the asm, whose stream ring stays in flight across compiler code.
// MODE 6 / 7: MODE 3 / 4 with the schedule table instead of the countdown + byte queue (round 4, the default);
// MODE 2: pairs of columns, sets prepared (rounds 2 - 3; SGL_TILED_PREP=1);  MODE 3: pairs of columns on half-set ring
// slots loaded with doubled lane rows, no preparation (round 4, the default for ranks 33 - 64);  MODE 4: quads of columns
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void acc_tiled_kernel(
    const uint32_t* __restrict__ sroff, const double* __restrict__ sx, const int64_t* __restrict__ cstart,
    const uint8_t* __restrict__ cnt, int T, int64_t nwb, const double* __restrict__ F, int k, int TR, int64_t nrow,
    int tiles_per_range, double* __restrict__ Bout, int64_t ncol, int KS, int ldf, int ldb, int64_t slab,
    const int32_t* __restrict__ perm, int range_fastest, const uint16_t* __restrict__ gtab, int tail_wg0, int tail_R,
    double* __restrict__ tail_part, int64_t tail_slab) {
    // k = factor rows handled by this launch (a part of the rank when it is above 64), KS = LDS row
    // stride the stream's offsets were built for, ldf / ldb = strides (doubles) between rows of F /
    // columns of the output, slab = doubles between the outputs of two tile ranges (blockIdx.y)
    constexpr int NSL = (MODE == 4 || MODE == 7) ? 4 : 2;
    constexpr bool PAIRRING = MODE == 3 || MODE == 6;
    asm volatile("" ::: "v255");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // work units in the order (column group, tile range), the range running fastest: the column groups come heaviest
    // first (sorted columns), so the longest units are dispatched first whatever the number of ranges
    // (only when the split was sized by that imbalance: otherwise the column groups run fastest, so that the
    // workgroups resident on one XCD stage the SAME factor tiles at about the same time and share them in its L2)
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (range_fastest) {
        const unsigned unit = blockIdx.x + gridDim.x * blockIdx.y, nranges = gridDim.y;
        bx = unit / nranges;
        by = unit - bx * nranges;
    }
    // tail split: the workgroups of the last, partly filled round of 256 come as tail_R pieces each (x = tail_wg0 + u)
    unsigned nry = gridDim.y;
    bool tail_piece = false;
    if (tail_R > 1 && (int)bx >= tail_wg0) {
        const unsigned u = bx - (unsigned)tail_wg0;
        bx = (unsigned)tail_wg0 + u / (unsigned)tail_R;
        by = u % (unsigned)tail_R;
        nry = (unsigned)tail_R;
        tail_piece = true;
    }
    const int64_t wb = (int64_t)bx * TILED_NW + wave;
    // tile range `by` of nry: sizes floor / ceil of T / ranges (tiles_per_range = the ceil, kept for the layout query)
    const int t0 = (int)(((int64_t)by * T) / nry);
    const int t1 = (int)((((int64_t)by + 1) * T) / nry);
    const bool wact = wb < nwb;
    typedef __attribute__((address_space(3))) char lds_char;
    // LDS byte address of this lane's pair of factor rows inside a tile row (NSL = 4: 16 lanes cover a column)
    constexpr int LMASK = NSL == 4 ? 15 : 31;
    const unsigned lane16 = (unsigned)(uintptr_t)(lds_char*)smem + (lane & LMASK) * 16;
    // stream slot this lane loads of a (half) set.  MODE 3: lane rows doubled -- lanes 0-15 and 16-31 the A half's 16
    // entries, lanes 32-47 and 48-63 the B half's (slots 32 ..): the [A A B B] layout the row broadcasts read
    const unsigned slot = PAIRRING ? (unsigned)((lane >> 5) * 32 + (lane & 15)) : (unsigned)lane;
    const unsigned voff4 = slot * 4, voff8 = slot * 8;
    // L2 prefetch of the stream (gen_acc_tiled.py, GenTab.wrap): lane offset into the row-offset stream that spreads the 64
    // lanes over one lap of it (8 ring slots: 1 KB in the pair layout, 2 KB in the quad layout), ACC_TILED_PF_LAPS laps ahead
    constexpr unsigned LAP_R = (NSL == 4 ? 256u : 128u) * 8u;
    const unsigned pfl = (unsigned)lane * (LAP_R / 64u) + (unsigned)ACC_TILED_PF_LAPS * LAP_R;

    asm volatile(ACC_TILED_ZERO_ASM ::: "memory");

    uint64_t rp = 0, xp = 0;   // next 64-entry set to LOAD (row offsets / values)
    int phase = 0;
    // group counts of the next chunk (32 bytes = one per column pair) and its size, read on the scalar side one
    // tile ahead: wb is wave-uniform
    uint64_t qn0 = 0, qn1 = 0, qn2 = 0, qn3 = 0;
    int64_t pos_next = 0;
    if (wact) {
        const int64_t pos = cstart[wb * T + t0];
        pos_next = cstart[wb * T + t0 + 1];
        const uint64_t* cq = reinterpret_cast<const uint64_t*>(cnt + (wb * T + t0) * TILED_NP);
        qn0 = cq[0]; qn1 = cq[1]; qn2 = cq[2]; qn3 = cq[3];
        const uint64_t r_ = reinterpret_cast<uint64_t>(sroff + pos), x_ = reinterpret_cast<uint64_t>(sx + pos);
        rp = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(r_ >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((unsigned)r_);
        xp = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(x_ >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((unsigned)x_);
        if constexpr (NSL == 4)
            asm volatile(ACC_TILED4_RING_FILL_ASM : [rp] "+s"(rp), [xp] "+s"(xp) : [voff4] "v"(voff4), [voff8] "v"(voff8) : ACC_TILED_CLOBBERS);
        else if constexpr (PAIRRING)
            asm volatile(ACC_TILED2R_RING_FILL_ASM : [rp] "+s"(rp), [xp] "+s"(xp) : [voff4] "v"(voff4), [voff8] "v"(voff8) : ACC_TILED_CLOBBERS);
        else
            asm volatile(ACC_TILED_RING_FILL_ASM : [rp] "+s"(rp), [xp] "+s"(xp) : [voff4] "v"(voff4), [voff8] "v"(voff8) : ACC_TILED_CLOBBERS);
    }
    if constexpr (NSL == 4) {
        // the tile rows are 256 B apart whatever the rank: bytes k * 8 .. 255 of a row are never staged; clear them
        // once (they only ever reach accumulator lanes that are not written out, but must stay finite: 0 * NaN)
        for (int e = (int)threadIdx.x; e < TILED_LDS_BYTES / 8; e += 64 * TILED_NW) tile[e] = 0.0;
    }

    int64_t pos_cur = wact ? cstart[wb * T + t0] : 0;
    for (int t = t0; t < t1; ++t) {
        const uint64_t q0 = qn0, q1 = qn1, q2 = qn2, q3 = qn3;
        const int nsets = (int)((pos_next - pos_cur) >> 6);  // 64-entry sets (32 per half) of this chunk
        const int64_t pos_chunk = pos_cur;                   // its first entry
        pos_cur = pos_next;
        if (wact && t + 1 < t1) {
            const uint64_t* cq = reinterpret_cast<const uint64_t*>(cnt + (wb * T + t + 1) * TILED_NP);
            qn0 = cq[0]; qn1 = cq[1]; qn2 = cq[2]; qn3 = cq[3];
            pos_next = cstart[wb * T + t + 2];
        }

        // stage rows [t*TR, ...) of F into LDS, row stride KS doubles
        const int64_t row0 = (int64_t)t * TR;
        const int rows = (int)((nrow - row0 < TR) ? (nrow - row0) : TR);
        const int n = rows * k;
        const double* __restrict__ src = F + row0 * ldf;
        if constexpr (NSL == 4) {
            // Quad layout: LDS rows of 32 doubles.  16 lanes per tile row (lane c moves the 16-byte piece c, or -- odd
            // rank / unaligned rows -- the doubles c and c + 16), 32 rows per round of the workgroup: no divisions, every
            // row's k * 8 contiguous bytes read by neighbouring lanes.  First round's loads before the barrier.
            const int c16 = (int)threadIdx.x & 15, r0 = (int)threadIdx.x >> 4;
            constexpr int RPR = 64 * TILED_NW / 16;   // rows per round
            const bool vec = ((k | ldf) & 1) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
            if (vec) {
                constexpr int RST = 4;
                double2 stg[RST];
                const bool cact = 2 * c16 < k;
#pragma unroll
                for (int j = 0; j < RST; ++j) {
                    const int r = r0 + j * RPR;
                    stg[j] = double2{0.0, 0.0};
                    if (cact && r < rows) stg[j] = *reinterpret_cast<const double2*>(src + (int64_t)r * ldf + 2 * c16);
                }
                __syncthreads();  // everyone is done reading the previous tile
                for (int rb = 0; rb < rows; rb += RST * RPR) {
#pragma unroll
                    for (int j = 0; j < RST; ++j) {
                        const int r = rb + r0 + j * RPR;
                        if (cact && r < rows) *reinterpret_cast<double2*>(tile + r * 32 + 2 * c16) = stg[j];
                    }
                    if (rb + RST * RPR < rows) {
#pragma unroll
                        for (int j = 0; j < RST; ++j) {
                            const int r = rb + RST * RPR + r0 + j * RPR;
                            stg[j] = double2{0.0, 0.0};
                            if (cact && r < rows) stg[j] = *reinterpret_cast<const double2*>(src + (int64_t)r * ldf + 2 * c16);
                        }
                    }
                }
            } else {
                __syncthreads();
                for (int r = r0; r < rows; r += RPR) {
                    if (c16 < k) tile[r * 32 + c16] = src[(int64_t)r * ldf + c16];
                    if (c16 + 16 < k) tile[r * 32 + c16 + 16] = src[(int64_t)r * ldf + c16 + 16];
                }
            }
        } else if (KS == k && ldf == k) {
            // Each thread moves up to NRND * RST 16-byte pieces of the (contiguous) tile in NRND rounds of
            // RST loads in flight.  The loads of the first round are issued before the barrier: they overlap
            // the tail of the previous tile's work of the other waves.
            constexpr int RST = 8;
            constexpr int NRND = (TILED_LDS_BYTES / 16 + RST * 64 * TILED_NW - 1) / (RST * 64 * TILED_NW);
            static_assert(NRND * RST * 64 * TILED_NW * 16 >= TILED_LDS_BYTES, "staging must cover the whole tile");
            double2 stg[RST];
#pragma unroll
            for (int j = 0; j < RST; ++j) {
                const int e = ((int)threadIdx.x + j * 64 * TILED_NW) * 2;
                stg[j] = double2{0.0, 0.0};
                if (e < n) stg[j] = *reinterpret_cast<const double2*>(src + e);
            }
            __syncthreads();  // everyone is done reading the previous tile
#pragma unroll
            for (int rd = 0; rd < NRND; ++rd) {
#pragma unroll
                for (int j = 0; j < RST; ++j) {
                    const int e = ((int)threadIdx.x + (rd * RST + j) * 64 * TILED_NW) * 2;
                    if (e < n) *reinterpret_cast<double2*>(tile + e) = stg[j];
                }
                if (rd + 1 < NRND) {
#pragma unroll
                    for (int j = 0; j < RST; ++j) {
                        const int e = ((int)threadIdx.x + ((rd + 1) * RST + j) * 64 * TILED_NW) * 2;
                        stg[j] = double2{0.0, 0.0};
                        if (e < n) stg[j] = *reinterpret_cast<const double2*>(src + e);
                    }
                }
            }
        } else {
            // odd k: rows are re-pitched to KS = k + 1 doubles (the pad column is never summed into a
            // stored factor row: lane 2l+1 == k is not written out)
            __syncthreads();
            // (also the path of a rank split into parts: rows of F are then ldf > k apart)
            if (((k | ldf) & 1) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
                const int hp = k >> 1;  // 16-byte pieces per row
                for (int e = (int)threadIdx.x; e < rows * hp; e += 64 * TILED_NW) {
                    const int r = e / hp, c2 = (e - r * hp) * 2;
                    *reinterpret_cast<double2*>(tile + r * KS + c2) = *reinterpret_cast<const double2*>(src + (int64_t)r * ldf + c2);
                }
            } else {
                for (int e = (int)threadIdx.x; e < n; e += 64 * TILED_NW) {
                    const int r = e / k, f = e - r * k;
                    tile[r * KS + f] = src[(int64_t)r * ldf + f];
                }
            }
            if (KS != k)
                for (int r = (int)threadIdx.x; r < rows; r += 64 * TILED_NW) tile[r * KS + k] = 0.0;
        }
        // all staging loads (and the ring loads in front of them) have landed: the asm's counted vmcnt waits see
        // only its own eight loads in flight
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
        if (wact && nsets > 0) {
            if constexpr (MODE == 6 || MODE == 7) {
                // first schedule word of this chunk (wave-uniform address)
                const uint64_t t_ = reinterpret_cast<uint64_t>(gtab + pos_chunk / (4 * NSL));
                const uint64_t tp = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(t_ >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((unsigned)t_);
                if constexpr (MODE == 6)
                    asm volatile(ACC_TILED2T_CHUNK_ASM
                                 : [rp] "+s"(rp), [xp] "+s"(xp), [phase] "+s"(phase)
                                 : [ns] "s"(2 * nsets), [tp] "s"(tp), [lane16] "v"(lane16), [voff4] "v"(voff4), [voff8] "v"(voff8), [pfl] "v"(pfl)
                                 : ACC_TILEDT_CLOBBERS);
                else
                    asm volatile(ACC_TILED4T_CHUNK_ASM
                                 : [rp] "+s"(rp), [xp] "+s"(xp), [phase] "+s"(phase)
                                 : [ns] "s"(nsets), [tp] "s"(tp), [lane16] "v"(lane16), [voff4] "v"(voff4), [voff8] "v"(voff8), [pfl] "v"(pfl)
                                 : ACC_TILEDT_CLOBBERS);
            } else if constexpr (MODE == 4)
                asm volatile(ACC_TILED4_CHUNK_ASM
                             : [rp] "+s"(rp), [xp] "+s"(xp), [phase] "+s"(phase)
                             : [ns] "s"(nsets), [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [q3] "s"(q3), [lane16] "v"(lane16),
                               [voff4] "v"(voff4), [voff8] "v"(voff8)
                             : ACC_TILED_CLOBBERS);
            else if constexpr (MODE == 3)
                asm volatile(ACC_TILED2R_CHUNK_ASM
                             : [rp] "+s"(rp), [xp] "+s"(xp), [phase] "+s"(phase)
                             : [ns] "s"(2 * nsets), [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [q3] "s"(q3), [lane16] "v"(lane16),
                               [voff4] "v"(voff4), [voff8] "v"(voff8)
                             : ACC_TILED_CLOBBERS);
            else
                asm volatile(ACC_TILED_CHUNK_ASM
                             : [rp] "+s"(rp), [xp] "+s"(xp), [phase] "+s"(phase)
                             : [ns] "s"(nsets), [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2), [q3] "s"(q3), [lane16] "v"(lane16),
                               [voff4] "v"(voff4), [voff8] "v"(voff8)
                             : ACC_TILED_CLOBBERS);
        }
    }
    // Up to four (eight) refills (issued past the end of this wave's range, into the stream's slack) are still in
    // flight; they write the ring registers v64..v75 (v64..v87) only, which nothing below touches
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wact) {
        double* out = Bout + (size_t)by * (size_t)slab;
        const int f = 2 * (lane & LMASK);
        const int64_t tail_pos0 = (int64_t)tail_wg0 * TILED_NW * (TILED_NP * NSL);
        for (int p = 0; p < TILED_NP; ++p) {
            double v0, v1;
            acc_load(4 * p, v0, v1);
            const int h = NSL == 4 ? (lane >> 4) : (lane >> 5);
            const int64_t col = tiled_slot_col(perm, ncol, wb, h, p, NSL);
            if (col >= 0) {
                if (tail_piece) {   // compact slab of this piece, indexed by the column's position in the stream's order
                    const int64_t pos = wb * (TILED_NP * NSL) + NSL * p + h;
                    double* o = tail_part + (size_t)by * (size_t)tail_slab + (pos - tail_pos0) * k;
                    if (f < k) o[f] = v0;
                    if (f + 1 < k) o[f + 1] = v1;
                } else {
                    if (f < k) out[col * ldb + f] = v0;
                    if (f + 1 < k) out[col * ldb + f + 1] = v1;
                }
            }
        }
    }
}

// B[col * ldb + f] = sum over the R tile-range slabs (compact k x ncol each, fixed order)
__global__ void acc_tiled_reduce_kernel(const double* __restrict__ part, int R, int64_t n, int k, int ldb, double* __restrict__ B) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        double s = part[e];
        for (int r = 1; r < R; ++r) s += part[(size_t)r * n + e];
        if (ldb == k) {
            B[e] = s;
        } else {
            const int64_t col = e / k;
            B[col * ldb + (e - col * k)] = s;
        }
    }
}

// tail split: B[col(pos) * ldb + f] = sum over the pieces' compact slabs, in piece order, for the stream positions pos >= pos0
__global__ void acc_tiled_reduce_tail_kernel(const double* __restrict__ part, int R, int64_t slab, int64_t pos0, int64_t ncol,
                                             const int32_t* __restrict__ perm, int k, int ldb, double* __restrict__ B) {
    const int64_t n = (ncol - pos0) * k;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        double s = part[e];
        for (int r = 1; r < R; ++r) s += part[(size_t)r * slab + e];
        const int64_t q = e / k;
        const int64_t col = perm ? (int64_t)perm[pos0 + q] : pos0 + q;
        B[col * ldb + (e - q * k)] = s;
    }
}

// F: first factor row of this part (row stride ldf), B: its first output row (column stride ldb), kf <= S.k
// factor rows.  One launch = one pass over the stream.
int k_acc_tiled(hipStream_t s, const DevTiled& S, const double* F, int ldf, double* B, int ldb, int kf, const double* xvals) {
    if (S.ncol <= 0) return SGL_OK;
    if (kf <= 0 || kf > S.k) { sgl_set_error("k_acc_tiled: bad part size %d (stream built for %d)", kf, S.k); return SGL_EINVAL; }
    const int KS = S.KS;
    const size_t lds = S.NSL == 4 ? (size_t)TILED_LDS_BYTES + 512 : (size_t)S.TR * KS * 8 + 512;
    // the attribute belongs to the (function, device) pair: one process may drive several devices
    static std::atomic<bool> attr_set[64];   // several host threads may drive devices at once (replica sweep)
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&acc_tiled_kernel<2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, TILED_LDS_BYTES + 512));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&acc_tiled_kernel<4>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, TILED_LDS_BYTES + 512));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&acc_tiled_kernel<3>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, TILED_LDS_BYTES + 512));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&acc_tiled_kernel<6>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, TILED_LDS_BYTES + 512));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&acc_tiled_kernel<7>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, TILED_LDS_BYTES + 512));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int64_t nwg_x = (S.nwb + TILED_NW - 1) / TILED_NW;
    const bool slabs = S.R > 1;
    double* out = slabs ? S.part : B;
    const int64_t n = (int64_t)kf * S.ncol;
    const bool tail = !slabs && S.tail_R > 1 && S.tail_wg0 >= 0;
    const int64_t tail_cols = tail ? (nwg_x - S.tail_wg0) * TILED_NW * S.CW : 0;
    const int64_t tail_slab = tail_cols * kf;
    const dim3 grid(tail ? (unsigned)(S.tail_wg0 + (nwg_x - S.tail_wg0) * S.tail_R) : (unsigned)nwg_x, (unsigned)S.R);
    const bool table = !getenv("SGL_TILED_NO_TABLE") && !getenv("SGL_TILED_PREP") && S.gtab != nullptr;
    auto launch = [&](auto mode) {
        constexpr int MODE = decltype(mode)::value;
        acc_tiled_kernel<MODE><<<grid, dim3(64 * TILED_NW), lds, s>>>(
            S.roff, xvals ? xvals : S.x, S.cstart, S.cnt, S.T, S.nwb, F, kf, S.TR, S.nrow, S.tiles_per_range, out, S.ncol, KS, ldf,
            slabs ? kf : ldb, slabs ? n : 0, S.perm, S.range_fastest ? 1 : 0, S.gtab, tail ? (int)S.tail_wg0 : -1, tail ? S.tail_R : 1,
            S.part, tail_slab);
    };
    if (S.NSL == 4 && table) launch(std::integral_constant<int, 7>());
    else if (S.NSL == 2 && table) launch(std::integral_constant<int, 6>());
    else if (S.NSL == 4) launch(std::integral_constant<int, 4>());
    else if (!getenv("SGL_TILED_PREP")) launch(std::integral_constant<int, 3>());
    else launch(std::integral_constant<int, 2>());
    HIPCHK(hipGetLastError());
    if (tail) {
        const int64_t pos0 = S.tail_wg0 * TILED_NW * S.CW;
        const int64_t ne = (S.ncol - pos0) * kf;
        if (ne > 0) {
            int64_t blocks = (ne + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            acc_tiled_reduce_tail_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(S.part, S.tail_R, tail_slab, pos0, S.ncol, S.perm, kf, ldb, B);
            HIPCHK(hipGetLastError());
        }
    }
    if (slabs) {
        int64_t blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        acc_tiled_reduce_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(S.part, S.R, n, kf, ldb, B);
        HIPCHK(hipGetLastError());
    }
    return SGL_OK;
}

// the whole rank: one pass per part of at most S.k factor rows (one part for k <= 64)
int k_acc_tiled_all(hipStream_t s, const DevTiled& S, const double* F, double* B, int k, const double* xvals) {
    for (int f0 = 0; f0 < k; f0 += S.k) {
        const int kf = (k - f0 < S.k) ? (k - f0) : S.k;
        SGLCHK(k_acc_tiled(s, S, F + f0, k, B + f0, k, kf, xvals));
    }
    return SGL_OK;
}

// part size the entry streams are built for at rank k (0: no tiled path): even, at most 64.  A pass costs the same whatever its
// part size within a layout (one address add, one ds_read_b128, two FMAs per entry tuple), and the quad layout of parts up to 32
// serves FOUR columns per tuple where the pair layout serves two: ranks 65 - 96 run as THREE quad passes (1.5 x the cost of a
// k = 64 pass) instead of two pair passes (2 x); 97 - 128 as FOUR quad passes (2.0 x the k = 64 pass; the two pair passes measure
// 2.4 x -- rhs_h per 200 000 cells at k = 100: 5.84 -> 4.89 ms, k = 128: 6.48 -> 5.01; profiles/r5_k_sweep_200k_cells.txt).
// One pass over half-height tiles with four factors per lane was priced and loses: 15 % more LDS time than the two passes
// (scripts/r5/r5_pad_model.py).  SGL_TILED_NO_QUAD3=1: two pair passes (A/B, tests).
// Round 6: ranks above 128 too -- ceil(k / 32) quad passes over the same stream (the reference's nnls / predict have no rank
// limit, src/singlet.cpp:229-250, and RunNMF hands ard_nmf k_max = 1e4, R/RunNMF.R:131); until then they fell to the plain
// wave-per-column gather (acc_kernel).  SGL_TILED_MAX_K=128 restores that (A/B, tests).
int tiled_part_size(int k) {
    if (k <= 64) return k;
    const char* mk = getenv("SGL_TILED_MAX_K");
    if (k > ((mk && atoi(mk) > 0) ? atoi(mk) : SGL_MAX_K)) return 0;
    const bool quad = !getenv("SGL_TILED_NO_QUAD3") && !getenv("SGL_TILED_NO_QUAD");
    const int parts = quad ? (k <= 96 ? 3 : (k + 31) / 32) : (k + 63) / 64;
    return ((k + parts - 1) / parts + 1) & ~1;
}
