// Masked (cross-validation) extras of c_ard_nmf:
//  * per-column Gram downdate a_i = a - AAt(submat(w, idx)) of predict_mask
//    (src/singlet.cpp:458-463, submat :211-216, AAt :200-206), where idx is the
//    set of rows j with draw(cell, gene) true for this column;
//  * mse_test (src/singlet.cpp:536-568).
// The hash is the bit-exact uint64 rng of kernels_hash.hip.
#include "sgl_internal.h"
#include "nnls_static_for.h"
#include <atomic>
#include <hipcub/hipcub.hpp>
#include <cstdlib>
#include <utility>
#include <type_traits>

// lower-triangle pair p -> (i, j), i >= j, p = i*(i+1)/2 + j
__device__ __forceinline__ void tri_unrank(int p, int& i, int& j) {
    int ii = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while ((ii + 1) * (ii + 2) / 2 <= p) ++ii;
    while (ii * (ii + 1) / 2 > p) --ii;
    i = ii;
    j = p - ii * (ii + 1) / 2;
}

// One workgroup (256 threads) per column.  Gout[c] = G - (Gsub + 1e-15 I).
template <int PMAX>
__global__ __launch_bounds__(256) void mask_gram_kernel(int64_t col0, int64_t ncols, int32_t nrow,
                                                        const int64_t* __restrict__ col_nnz,
                                                        const double* __restrict__ F, const double* __restrict__ G,
                                                        int k, uint64_t seed, SglDiv inv_density, int mask_t,
                                                        int64_t col_off, int64_t row_off, double* __restrict__ Gout, int raw,
                                                        int pair0) {
    // pair0: first lower-triangle pair of this launch (ranks whose triangle exceeds 256 * PMAX pairs take several
    // launches, each hashing the rows again: the slow, any-rank path)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    int* list = reinterpret_cast<int*>(smem_raw);                 // [256] drawn rows of the current chunk
    int* wcount = list + 256;                                      // [4] per-wave counts (+pad to 8 ints)
    double* tile = reinterpret_cast<double*>(smem_raw + 264 * 4);  // [16][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t lc = blockIdx.x;  // local column in this launch
    if (lc >= ncols) return;
    const int64_t col = col0 + lc;
    double* out = Gout + (size_t)lc * k * k;
    if (col_nnz != nullptr && col_nnz[col] == 0) return;  // column is skipped by predict_mask (l.444)

    const int npairs = k * (k + 1) / 2;
    int pi[PMAX], pj[PMAX];
    double acc[PMAX];
#pragma unroll
    for (int q = 0; q < PMAX; ++q) {
        const int p = pair0 + tid + 256 * q;
        pi[q] = 0; pj[q] = 0; acc[q] = 0.0;
        if (p < npairs) tri_unrank(p, pi[q], pj[q]);
    }
    const uint64_t gcol = (uint64_t)(col + col_off);
    for (int64_t r0 = 0; r0 < nrow; r0 += 256) {
        const int64_t r = r0 + tid;
        bool drawn = false;
        if (r < nrow) {
            const uint64_t grow = (uint64_t)(r + row_off);
            drawn = mask_t ? sgl_draw(seed, grow, gcol, inv_density) : sgl_draw(seed, gcol, grow, inv_density);
        }
        const unsigned long long m = __ballot(drawn);
        __syncthreads();  // previous chunk's list fully consumed
        if (lane == 0) wcount[wave] = __popcll(m);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wcount[w];
        const int total = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        if (drawn) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)r;
        __syncthreads();
        for (int t0 = 0; t0 < total; t0 += 16) {
            const int nb = (total - t0 < 16) ? (total - t0) : 16;
            for (int e = tid; e < nb * k; e += 256) {
                const int t = e / k, f = e - t * k;
                tile[t * k + f] = F[(int64_t)list[t0 + t] * k + f];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < PMAX; ++q) {
                if (pair0 + tid + 256 * q < npairs) {
                    double a = acc[q];
                    for (int t = 0; t < nb; ++t) a = fma(tile[t * k + pi[q]], tile[t * k + pj[q]], a);
                    acc[q] = a;
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int q = 0; q < PMAX; ++q) {
        if (pair0 + tid + 256 * q < npairs) {
            const int i = pi[q], j = pj[q];
            double sub = acc[q];
            if (i == j && !raw) sub += 1e-15;  // AAt(wsub) adds it too (quirk 8)
            const double v = raw ? sub : G[(size_t)j * k + i] - sub;
            out[(size_t)j * k + i] = v;
            out[(size_t)i * k + j] = v;
        }
    }
}

// Same downdate on the FP64 matrix cores, k <= 128 (NT = ceil(k / 16) <= 8).  AAt(submat(w, idx)) is a
// k x |idx| by |idx| x k contraction: with v_mfma_f64_16x16x4_f64 (operand layout in kernels_dense.hip)
// four drawn rows feed all NT (NT + 1) / 2 lower-triangle 16 x 16 tiles from NT loads per lane, where
// the VALU kernel above reads two LDS operands per FMA.  One workgroup per column, every wave on its own
// quarter of the rows: per STEP a wave hashes 64 rows (one per lane; ~50 VALU instructions), appends the
// drawn ones to its ring queue in LDS, and feeds ONE queued group of 4 rows to the matrix cores.  Hash and
// MFMAs of a step sit in one basic block, interleaved by sched_group_barrier: a v_mfma_f64_16x16x4 occupies
// the matrix pipe for 64 cycles during which the wave issues the next hash instructions (round 1 hashed
// ALL its rows first and only then turned to the matrix cores: 46 % MFMA-busy; profiles/).  The operands
// of the next group are gathered (L2) while the current step runs.  The 4 waves' tiles are summed at the end.
typedef double mg_d4 __attribute__((ext_vector_type(4)));
template <typename F_, int... Is>
__device__ __forceinline__ void static_for_rem_impl(F_&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F_>
__device__ __forceinline__ void static_for_rem(F_&& f) { static_for_rem_impl(f, std::make_integer_sequence<int, (N > 0 ? N : 1)>{}); }
#define MG_QW 1024  // ring queue of drawn rows per wave (power of two)
// tiles per epilogue batch (8 KB of LDS each); at least what the remainder partials need (REM * NB * 2 KB)
constexpr int mg_tb(int ntiles, int rem, int nb) {
    // small tile sets run several workgroups per CU: keep their LDS small (17 KB of row queues + 8 KB per tile of the batch)
    const int want = ntiles <= 6 ? 2 : (ntiles <= 15 ? 4 : 8), need = ((rem < 4 ? rem : 4) * nb + 3) / 4;   // (the list kernel's remainder quads go through one at a time)
    return want > need ? want : need;
}
#define MG_TB(NTILES_, REM_, NB_) mg_tb(NTILES_, REM_, NB_)

// NPARTS > 1 (k > 96): the tile set no longer fits a wave's registers; launch PART = 0 .. NPARTS-1, each
// computing the tiles t with t % NPARTS == PART (the rows are hashed again in every part).
// REM > 0 (k = 16 NT + r with r <= REM, REM = 2 or 4): the last r factor rows would cost NT + 1 more, almost
// empty, 16 x 16 tiles (k = 50: 10 tiles where 6 are full).  They are updated on the VALU instead, in the
// MFMAs' shadow: lane (kk, r16) of a row group holds F[row_kk, 16 b + r16] for every block b, the factors
// 16 NT + i (i < REM) of that row sit in lanes r16 = i of block NT and reach the group's 16 lanes by DPP
// row_newbcast:i -- 2 REM (NT + 1) FP64 FMAs per group of 4 rows instead of NT + 1 MFMAs.
template <int J>
__device__ __forceinline__ double mg_bcast(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + J, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + J, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}


// ---- epilogue of the MFMA downdate kernels: sum the 4 waves' partial tiles, Gout = G - (Gsub + 1e-15 I) ----------
// Tiles go through LDS in batches of TB: ONE barrier pair per batch, then wave w finishes the batch's tiles
// u = w, w + 4, ... (sum of the four partials in wave order, as ever; the Gram value; the two mirrored stores), so
// that the loads and stores of all tiles are in flight together.  The first form took the tiles one by one -- two
// barriers, wave 0 alone reading G and storing -- ~1 us per tile with the matrix pipe idle: 55 us per column at
// k = 100, 30 % of the H-side launches (profiles/r3_mask_gram_ablation.md).  Same sums in the same order.
constexpr int mg_tile_bi(int t) { int bi = 0; while ((bi + 1) * (bi + 2) / 2 <= t) ++bi; return bi; }
constexpr int mg_tile_bj(int t) { return t - mg_tile_bi(t) * (mg_tile_bi(t) + 1) / 2; }

// tile(integral_constant<int, q>) returns the wave's partial of tile q of this part (a register array in the hashing
// kernel; read from the AGPR file batch by batch in the list kernel, which keeps its VGPR count low)
template <int NT, int NPARTS, int PART, int REM, int TB, int NTILES, typename TileFn, int NB, int NR>
__device__ __forceinline__ void mg_epilogue(TileFn&& tile, const double (&accr)[NR][NB], double* __restrict__ epi, int wave,
                                            int lane, const double* __restrict__ G, double* __restrict__ out, int k, int raw) {
    const int r16 = lane & 15, kk = lane >> 4;
    constexpr int NBATCH = (NTILES + TB - 1) / TB;
    static_for<NBATCH>([&](auto bc) {
        constexpr int q0 = decltype(bc)::value * TB;
        __syncthreads();   // the main loop (first batch) / the reads of the previous batch are done
        static_for<TB>([&](auto uc) {
            constexpr int u = decltype(uc)::value, qa = q0 + u;
            if constexpr (qa < NTILES) {
                double* dst = epi + (u * 4 + wave) * 256 + lane * 4;
                const mg_d4 a = tile(std::integral_constant<int, qa>{});
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r] = a[r];
            }
        });
        __syncthreads();
        static_for<TB>([&](auto uc) {
            constexpr int u = decltype(uc)::value, qa = q0 + u;
            if constexpr (qa < NTILES) {
                if ((u & 3) == wave) {
                    constexpr int t = PART + qa * NPARTS, bi = mg_tile_bi(t), bj = mg_tile_bj(t);
                    const double* src = epi + u * 4 * 256 + lane * 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double sub = ((src[r] + src[256 + r]) + src[512 + r]) + src[768 + r];
                        const int row = bi * 16 + kk + 4 * r;  // D row (lane >> 4) + 4 r
                        const int cc = bj * 16 + r16;          // D col lane & 15
                        if (row < k && cc < k) {
                            if (row == cc && !raw) sub += 1e-15;
                            const double v = raw ? sub : G[(size_t)cc * k + row] - sub;
                            out[(size_t)cc * k + row] = v;
                            if (row != cc) out[(size_t)row * k + cc] = v;
                        }
                    }
                }
            }
        });
    });
    if constexpr (REM > 0) {   // remainder rows: the 4 row groups of the 4 waves, fixed order (w, then g)
        static_assert(REM * NB * 4 * 64 <= TB * 4 * 256, "the remainder partials must fit the tile buffer");
        __syncthreads();
        static_for<REM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
#pragma unroll
            for (int b = 0; b < NB; ++b) epi[((i * NB + b) * 4 + wave) * 64 + lane] = accr[i][b];
        });
        __syncthreads();
        for (int o = wave * 64 + lane; o < REM * NB * 16; o += 256) {
            const int ib = o >> 4, c16 = o & 15, i = ib / NB, b = ib - i * NB;
            const double* src = epi + (size_t)ib * 4 * 64 + c16;
            double sub = 0.0;
            for (int w = 0; w < 4; ++w)
                for (int g = 0; g < 4; ++g) sub += src[w * 64 + g * 16];
            const int row = 16 * NT + i, cc = 16 * b + c16;
            if (row < k && cc < k && cc <= row) {
                if (row == cc && !raw) sub += 1e-15;
                const double v = raw ? sub : G[(size_t)cc * k + row] - sub;
                out[(size_t)cc * k + row] = v;
                if (row != cc) out[(size_t)row * k + cc] = v;
            }
        }
    }
}

template <int NT, int NPARTS = 1, int PART = 0, int REM = 0>
__global__ __launch_bounds__(256) void mask_gram_mfma_kernel(int64_t col0, int64_t ncols, int32_t nrow,
                                                             const int64_t* __restrict__ col_nnz,
                                                             const double* __restrict__ F, const double* __restrict__ G,
                                                             int k, uint64_t seed, SglDiv inv_density, int mask_t,
                                                             int64_t col_off, int64_t row_off, double* __restrict__ Gout, int raw) {
    constexpr int NTILES_ALL = NT * (NT + 1) / 2;
    constexpr int NTILES = (NTILES_ALL - PART + NPARTS - 1) / NPARTS;  // tiles of this part
    __shared__ int list[4 * (MG_QW + 64)];   // per wave: the ring + a dump slot for the lanes that drew nothing
    extern __shared__ __attribute__((aligned(16))) double mg_epi_raw[];   // epilogue buffer: MG_TB tiles x 4 waves x 256
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int64_t lc = blockIdx.x;
    if (lc >= ncols) return;
    const int64_t col = col0 + lc;
    double* out = Gout + (size_t)lc * k * k;
    if (col_nnz != nullptr && col_nnz[col] == 0) return;  // column is skipped by predict_mask (l.444)

    constexpr int NB = NT + (REM > 0 ? 1 : 0);   // factor blocks a lane loads per row
    static_assert(REM == 0 || NPARTS == 1, "the VALU remainder is built for single-part launches");
    mg_d4 acc[NTILES];
#pragma unroll
    for (int t = 0; t < NTILES; ++t) acc[t] = mg_d4{0, 0, 0, 0};
    double accr[REM > 0 ? REM : 1][NB];          // VALU remainder: rows 16 NT + i against columns 16 b + r16
#pragma unroll
    for (int i = 0; i < (REM > 0 ? REM : 1); ++i)
#pragma unroll
        for (int b = 0; b < NB; ++b) accr[i][b] = 0.0;
    int* q = list + wave * (MG_QW + 64);
    unsigned head = 0, tail = 0;   // rows queued / rows handed to the matrix cores (wave-uniform, monotone)
    // operands of one group for this lane: row q[tail + kk] (kk = lane / 16), factors b * 16 + r16; rows
    // beyond `avail` contribute zeros (the last, partial group)
    auto load_group = [&](double (&f)[NB], unsigned first, int avail) {
        const bool valid = kk < avail;
        const int row = valid ? q[(first + kk) & (MG_QW - 1)] : 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int fr = b * 16 + r16;
            f[b] = (valid && fr < k) ? F[(int64_t)row * k + fr] : 0.0;
        }
    };
    // the same for a complete group, branch-free: only the last block can reach past the rank
    auto load_group_full = [&](double (&f)[NB], unsigned first) {
        const double* src = F + (int64_t)q[(first + kk) & (MG_QW - 1)] * k + r16;
#pragma unroll
        for (int b = 0; b + 1 < NB; ++b) f[b] = src[16 * b];
        f[NB - 1] = ((NB - 1) * 16 + r16 < k) ? src[16 * (NB - 1)] : 0.0;
    };
    auto mfma_group = [&](const double (&f)[NB]) {
        if constexpr (REM > 0) {   // the remainder rows on the VALU
            static_for_rem<REM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const double v = mg_bcast<i>(f[NT]);
#pragma unroll
                for (int b = 0; b < NB; ++b) accr[i][b] = fma(f[b], v, accr[i][b]);
            });
        }
        int t = 0, qi = 0;
#pragma unroll
        for (int bi = 0; bi < NT; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) {
                if (t % NPARTS == PART) {
                    acc[qi] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[bi], f[bj], acc[qi], 0, 0, 0);
                    ++qi;
                }
                ++t;
            }
    };
    const uint64_t gcol = (uint64_t)(col + col_off);
    // branch-free hash step: lanes that drew nothing write to the dump slot
    auto hash_step = [&](int64_t r0) {
        const int64_t r = r0 + lane;
        bool drawn = false;
        if (r < nrow) {
            const uint64_t grow = (uint64_t)(r + row_off);
            drawn = mask_t ? sgl_draw(seed, grow, gcol, inv_density) : sgl_draw(seed, gcol, grow, inv_density);
        }
        const unsigned long long m = __ballot(drawn);
        const unsigned pos = drawn ? ((head + (unsigned)__popcll(m & ((1ull << lane) - 1ull))) & (MG_QW - 1)) : (unsigned)(MG_QW + lane);
        q[pos] = (int)r;
        head += (unsigned)__popcll(m);
    };

    double f[NB];
    bool have = false;
    for (int64_t r0 = (int64_t)wave * 64; r0 < nrow; r0 += 256) {
        // f holds the group gathered during the previous step (if any).  Copy it, start the gather of the NEXT group
        // at once -- it has this step's hash and MFMAs to arrive in -- then hash 64 rows and feed the copy to the
        // matrix cores.  (A group queued by this step's hash is gathered at the top of the next step.)
        double fc[NB];
        const bool have_c = have;
        if (have_c) {
#pragma unroll
            for (int b = 0; b < NB; ++b) fc[b] = f[b];
        }
        have = head - tail >= 4;
        if (have) {
            load_group_full(f, tail);
            tail += 4;
        }
        if (have_c) {
            hash_step(r0);
            mfma_group(fc);
#pragma unroll
            for (int t = 0; t < NTILES; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                               // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x2, (56 + 3 * REM * NB + NTILES - 1) / NTILES, 0);   // its share of the hash (+ remainder FMAs)
            }
        } else {
            hash_step(r0);
        }
        __builtin_amdgcn_wave_barrier();
        // the queue must never wrap onto unread rows: drain without hashing while it is nearly full (dense masks)
        while (head - tail > MG_QW - 128) {
            if (have) {   // the gathered group first: rows enter the sums in queue order
                mfma_group(f);
                have = false;
            }
            double fz[NB];
            load_group_full(fz, tail);
            tail += 4;
            mfma_group(fz);
        }
    }
    if (have) mfma_group(f);
    while (head != tail) {   // what is left, the last group possibly partial
        const int avail = (int)((head - tail < 4u) ? (head - tail) : 4u);
        double fz[NB];
        load_group(fz, tail, avail);
        tail += (unsigned)avail;
        mfma_group(fz);
    }

    mg_epilogue<NT, NPARTS, PART, REM, MG_TB(NTILES, REM, NB), NTILES>([&](auto qc) { return acc[decltype(qc)::value]; }, accr, mg_epi_raw, wave, lane, G,
                                                                       out, k, raw);
}

template <int NT, int NPARTS = 1, int PART = 0, int REM = 0>
static int launch_mask_gram_mfma(dim3 g, dim3 b, hipStream_t s, int64_t col0, int64_t ncols, int32_t nrow, const int64_t* col_nnz,
                                 const double* F, const double* G, int k, uint64_t seed, SglDiv dv, int mask_t, int64_t col_offset,
                                 int64_t row_offset, double* Gcols, int raw) {
    constexpr int NTILES = (NT * (NT + 1) / 2 - PART + NPARTS - 1) / NPARTS, NB = NT + (REM > 0 ? 1 : 0);
    constexpr size_t lds = (size_t)MG_TB(NTILES, REM, NB) * 4 * 256 * sizeof(double);
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_gram_mfma_kernel<NT, NPARTS, PART, REM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    mask_gram_mfma_kernel<NT, NPARTS, PART, REM><<<g, b, lds, s>>>(col0, ncols, nrow, col_nnz, F, G, k, seed, dv, mask_t, col_offset, row_offset, Gcols, raw);
    return SGL_OK;
}

#include "mask_gram_list.inc"

// G == nullptr: "raw" mode -- Gcols[c] = the plain sum of f f^T over the masked rows of column c (no Gram,
// no ridge): the partial a cell shard contributes to a gene's downdate (multi.hip sums them over the ranks)
int k_mask_gram_cols(hipStream_t s, int64_t col0, int64_t ncols, int32_t nrow, const int64_t* col_nnz,
                     const double* F, const double* G, int k, uint64_t seed, uint64_t inv_density, int mask_t,
                     int64_t col_offset, int64_t row_offset, double* Gcols, const DevMaskList* L) {
    if (ncols <= 0) return SGL_OK;
    const int raw = G == nullptr ? 1 : 0;
    dim3 g((unsigned)ncols), b(256);
    const bool lists = L != nullptr && L->mask_t == mask_t && L->ptr != nullptr && L->idx != nullptr && L->seed == seed &&
                       L->inv == inv_density && L->nrow == nrow && L->col_off == col_offset && L->row_off == row_offset &&
                       col0 + ncols <= L->ncol;
    if (k <= 128 && !getenv("SGL_MASK_GRAM_VALU")) {  // env: keep the VALU kernel reachable for A/B tests
#define SGL_MGM(...)                                                                                                               \
    do {                                                                                                                            \
        if (lists) SGLCHK((launch_mask_gram_list<__VA_ARGS__>(g, b, s, col0, ncols, col_nnz, L, F, G, k, Gcols, raw)));            \
        else SGLCHK((launch_mask_gram_mfma<__VA_ARGS__>(g, b, s, col0, ncols, nrow, col_nnz, F, G, k, seed, sgl_div_make(inv_density), mask_t, col_offset, row_offset, Gcols, raw))); \
    } while (0)
        // k = 16 NT + r with 4 < r <= 8 at NT = 3 .. 5, from the lists: the remainder rows as two quads of quarter-MFMAs instead
        // of NT + 1 more tiles.  A v_mfma_f64_4x4x4_4b costs ~22 cycles here, not a quarter of the 16x16x4's 64; measured
        // (mask phase per iteration at 30 000 x 200 000, ms): k = 56 48.3 -> 43.0, 69 70.0 -> 63.1, 72 70.2 -> 63.0, 88 97.7 ->
        // 90.6; it does not pay with three quads (k = 60 47.2 -> 47.3, 76 70.7 -> 72.9, 90 96.0 -> 95.1) or below NT = 3
        // (k = 24 18.5 -> 22.8, 40 29.9 -> 29.8, 44 34.1 -> 35.9): those keep the tile count below, and so does the hashing kernel
        const int nt_full = k / 16, rem = k % 16;
        // Round 6: the rows beyond 16 NT on the VALU (mask_gram_list_kernel<NT, 1, 0, 0, REMV>): REMV (NT + 1) v_fmac_f64_dpp per group
        // against (NT + 1) quarter-MFMAs of ~22 cycles per remainder quad or NT + 1 MFMAs of 64 for a further tile row.  Priced at
        // ~4.7 cycles per DPP FMA it should have won everywhere up to 12 rows; measured (mask phase per masked iteration at 30 000 x
        // 200 000, ms, before -> after; profiles/r6_mask_gram_valu_remainder_ab.txt) it wins where the rows are few and the tile set
        // small -- k = 20 15.9 -> 13.4, 34 23.8 -> 22.6, 40 27.0 -> 25.0, 50 34.3 -> 33.6, 66 50.3 -> 48.0 -- and loses or ties beyond:
        // k = 44 29.6 -> 29.7 (12 rows), 56 37.6 -> 38.9 (8 rows at NT = 3), 60 43.4 -> 47.0 (12), 70 58.4 -> 58.8 (6 at NT = 4): an FMA
        // on the remainder accumulators costs ~9 cycles here, not 4.7.  Kept: up to 4 rows at NT = 1, up to 8 at NT = 2, up to 2 at
        // NT = 3 and 4.  SGL_MASK_GRAM_NO_REMV=1: the forms below (A/B, tests).
        const int rv_max = nt_full == 1 ? 4 : (nt_full == 2 ? 8 : ((nt_full == 3 || nt_full == 4) ? 2 : 0));
        if (lists && rem > 0 && rem <= rv_max && !getenv("SGL_MASK_GRAM_NO_REM") && !getenv("SGL_MASK_GRAM_NO_REMV")) {
#define SGL_MGV(NT_, RV_) SGLCHK((launch_mask_gram_list<NT_, 1, 0, 0, RV_>(g, b, s, col0, ncols, col_nnz, L, F, G, k, Gcols, raw)))
            const int rv = rem <= 2 ? 2 : (rem <= 4 ? 4 : 8);
            switch (nt_full * 100 + rv) {
                case 102: SGL_MGV(1, 2); break;
                case 104: SGL_MGV(1, 4); break;
                case 202: SGL_MGV(2, 2); break;
                case 204: SGL_MGV(2, 4); break;
                case 208: SGL_MGV(2, 8); break;
                case 302: SGL_MGV(3, 2); break;
                default: SGL_MGV(4, 2); break;
            }
#undef SGL_MGV
            HIPCHK(hipGetLastError());
            return SGL_OK;
        }
        if (lists && rem > 4 && rem <= 8 && nt_full >= 3 && nt_full <= 5 && !getenv("SGL_MASK_GRAM_NO_REM") && !getenv("SGL_MASK_GRAM_NO_REM8")) {
#define SGL_MGL(NT_, REM_) SGLCHK((launch_mask_gram_list<NT_, 1, 0, REM_>(g, b, s, col0, ncols, col_nnz, L, F, G, k, Gcols, raw)))
            switch (nt_full) {
                case 3: SGL_MGL(3, 8); break;
                case 4: SGL_MGL(4, 8); break;
                default: SGL_MGL(5, 8); break;
            }
#undef SGL_MGL
            HIPCHK(hipGetLastError());
            return SGL_OK;
        }
        // k = 16 NT + r with r <= 4: the remainder rows as quarter-MFMAs (lists) / on the VALU (hashing kernel), REM = 2 / 4
        if (rem >= 1 && rem <= 4 && nt_full >= 1 && nt_full <= 6 && !getenv("SGL_MASK_GRAM_NO_REM")) {
            const int key = nt_full * 10 + (rem <= 2 ? 2 : 4);
            switch (key) {
                case 12: SGL_MGM(1, 1, 0, 2); break;
                case 14: SGL_MGM(1, 1, 0, 4); break;
                case 22: SGL_MGM(2, 1, 0, 2); break;
                case 24: SGL_MGM(2, 1, 0, 4); break;
                case 32: SGL_MGM(3, 1, 0, 2); break;
                case 34: SGL_MGM(3, 1, 0, 4); break;
                case 42: SGL_MGM(4, 1, 0, 2); break;
                case 44: SGL_MGM(4, 1, 0, 4); break;
                case 52: SGL_MGM(5, 1, 0, 2); break;
                case 54: SGL_MGM(5, 1, 0, 4); break;
                case 62: SGL_MGM(6, 1, 0, 2); break;
                default: SGL_MGM(6, 1, 0, 4); break;
            }
            HIPCHK(hipGetLastError());
            return SGL_OK;
        }
        switch ((k + 15) / 16) {
            case 1: SGL_MGM(1); break;
            case 2: SGL_MGM(2); break;
            case 3: SGL_MGM(3); break;
            case 4: SGL_MGM(4); break;
            case 5: SGL_MGM(5); break;
            case 6: SGL_MGM(6); break;
            case 7: SGL_MGM(7, 2, 0); SGL_MGM(7, 2, 1); break;
            default: SGL_MGM(8, 2, 0); SGL_MGM(8, 2, 1); break;
        }
#undef SGL_MGM
        HIPCHK(hipGetLastError());
        return SGL_OK;
    }
    const int npairs = k * (k + 1) / 2;
    const int P = (npairs + 255) / 256;
    const size_t smem = 264 * 4 + sizeof(double) * 16 * (size_t)k;
#define SGL_MG(PM, P0) mask_gram_kernel<PM><<<g, b, smem, s>>>(col0, ncols, nrow, col_nnz, F, G, k, seed, sgl_div_make(inv_density), mask_t, col_offset, row_offset, Gcols, raw, P0)
    if (P <= 4) SGL_MG(4, 0);
    else if (P <= 9) SGL_MG(9, 0);
    else if (P <= 20) SGL_MG(20, 0);
    else {
        if (smem > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_gram_kernel<33>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        for (int p0 = 0; p0 < npairs; p0 += 256 * 33) SGL_MG(33, p0);   // k > 128: one launch per 8448 pairs of the triangle
    }
#undef SGL_MG
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// a_i = G - (S_i + 1e-15 I) for ncols columns from their summed downdates S (src/singlet.cpp:460-462:
// AAt(wsub) carries the 1e-15 ridge too, so it cancels the one inside G)
__global__ void mask_gram_finalize_kernel(const double* __restrict__ G, const double* __restrict__ S, int k, int64_t ncols,
                                          double* __restrict__ out) {
    const int64_t kk = (int64_t)k * k, n = kk * ncols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(e % kk);
        double sub = S[e];
        if (q / k == q % k) sub += 1e-15;
        out[e] = G[q] - sub;
    }
}

int k_mask_gram_finalize(hipStream_t s, const double* G, const double* S, int k, int64_t ncols, double* out) {
    if (ncols <= 0) return SGL_OK;
    const int64_t n = (int64_t)k * k * ncols;
    mask_gram_finalize_kernel<<<dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, s>>>(G, S, k, ncols, out);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// The team's per-gene downdates travel as lower triangles (round 6): S_g is symmetric, so the reduce-scatter of the masked
// W-update (multi.hip: k x k x genes doubles per rank and iteration, 2.4 GB at k = 100) moves k (k + 1) / 2 per gene instead of
// k^2.  tri[c][i (i + 1) / 2 + j] = S[c][j * k + i] for j <= i; the sums over the ranks are element-wise either way: same bits.
__global__ void tri_pack_kernel(const double* __restrict__ S, int k, int64_t ncols, double* __restrict__ tri) {
    const int64_t T = (int64_t)k * (k + 1) / 2, n = T * ncols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = e / T;
        const int t = (int)(e - c * T);
        int i = (int)((__builtin_sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;   // (guards the rounding of the square root)
        while (i * (i + 1) / 2 > t) --i;
        const int j = t - i * (i + 1) / 2;
        tri[e] = S[c * (int64_t)k * k + (int64_t)j * k + i];
    }
}
int k_tri_pack(hipStream_t s, const double* S, int k, int64_t ncols, double* tri) {
    if (ncols <= 0) return SGL_OK;
    const int64_t n = (int64_t)k * (k + 1) / 2 * ncols;
    tri_pack_kernel<<<dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, s>>>(S, k, ncols, tri);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}
// mask_gram_finalize_kernel from packed triangles
__global__ void mask_gram_finalize_tri_kernel(const double* __restrict__ G, const double* __restrict__ tri, int k, int64_t ncols,
                                              double* __restrict__ out) {
    const int64_t kk = (int64_t)k * k, n = kk * ncols, T = (int64_t)k * (k + 1) / 2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = e / kk;
        const int q = (int)(e - c * kk);
        const int a = q / k, b = q % k, i = a > b ? a : b, j = a > b ? b : a;
        double sub = tri[c * T + (int64_t)i * (i + 1) / 2 + j];
        if (a == b) sub += 1e-15;
        out[e] = G[q] - sub;
    }
}
int k_mask_gram_finalize_tri(hipStream_t s, const double* G, const double* tri, int k, int64_t ncols, double* out) {
    if (ncols <= 0) return SGL_OK;
    const int64_t n = (int64_t)k * k * ncols;
    mask_gram_finalize_tri_kernel<<<dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, s>>>(G, tri, k, ncols, out);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// ---------------------------------------------------------------- mse_test --
// mse_test (src/singlet.cpp:536-568).  One wave per cell, LANES OVER THE FACTORS: h[:, cell] sits in
// registers (one or two per lane), every 64-gene chunk is hashed one gene per lane, and for each drawn gene
// (wave-uniform loop over the ballot) the wave reads Wd[:, gene] as ONE coalesced k * 8-byte line, multiplies
// and sums over the lanes with DPP row operations.  The matrix value at (gene, cell) comes from a window of
// the cell's (ascending) non-zeros that slides along with the gene chunks: the entries of the current chunk
// are scattered into a 64-slot LDS row, lane b picks up slot b, and the drawn gene at bit b takes it by
// v_readlane.  (Round 1 had one drawn gene per lane: every lane walked its own 400-byte row of Wd and did
// its own binary search -- 35 ms per call at 30 000 x 100 000, k = 50; this form: see profiles/.)
// The k-long dot is summed as a lane tree instead of left to right (rounding only: ~1e-16 relative).
__device__ __forceinline__ double dpp_mov_f64(double v, const int ctrl, const int row_mask) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    int rl, rh;
    switch (ctrl) {   // the control word must be an immediate
        case 0: rl = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false); break;   // quad_perm [1,0,3,2]
        case 1: rl = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, false); break;   // quad_perm [2,3,0,1]
        case 2: rl = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, false); break; // row_half_mirror
        case 3: rl = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, false); break; // row_mirror
        case 4: rl = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xa, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xa, 0xf, false); break; // row_bcast:15 into rows 1, 3
        default: rl = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xc, 0xf, false); rh = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xc, 0xf, false); break; // row_bcast:31 into rows 2, 3
    }
    (void)row_mask;
    return __hiloint2double(rh, rl);
}
// sum over the 64 lanes, returned wave-uniform
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_mov_f64(v, 0, 0xf);
    v += dpp_mov_f64(v, 1, 0xf);
    v += dpp_mov_f64(v, 2, 0xf);
    v += dpp_mov_f64(v, 3, 0xf);   // every lane of a 16-lane row holds its row's sum
    v += dpp_mov_f64(v, 4, 0xa);   // rows 1 and 3 add the sum of the row before
    v += dpp_mov_f64(v, 5, 0xc);   // rows 2 and 3 add lane 31: lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

template <int R>
__global__ __launch_bounds__(256) void mse_test_kernel(const double* __restrict__ Ax, const int32_t* __restrict__ Ai,
                                                       const int64_t* __restrict__ Ap, int32_t m, int64_t n,
                                                       int64_t cell_off, const double* __restrict__ Wd,
                                                       const double* __restrict__ H, int k, uint64_t seed,
                                                       SglDiv inv_density, double* __restrict__ losses) {
    __shared__ double slot[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    double* sl = slot[wave];
    for (int64_t cell = gw; cell < n; cell += nwaves) {
        const uint64_t xi = sgl_rand_i(seed, (uint64_t)(cell + cell_off));
        const int64_t qend = Ap[cell + 1];
        int64_t qbase = Ap[cell];
        double h[R];
#pragma unroll
        for (int r = 0; r < R; ++r) h[r] = (lane + 64 * r < k) ? H[cell * k + lane + 64 * r] : 0.0;
        // window of the cell's non-zeros: entry qbase + lane
        int wrow = (qbase + lane < qend) ? Ai[qbase + lane] : INT32_MAX;
        double wval = (qbase + lane < qend) ? Ax[qbase + lane] : 0.0;
        double s = 0.0;
        long long cnt = 0;
        for (int g0 = 0; g0 < m; g0 += 64) {
            const int g = g0 + lane;
            const bool drawn = (g < m) && sgl_divides(sgl_rand_j(xi, (uint64_t)g), inv_density);
            unsigned long long mk = __ballot(drawn);
            if (mk == 0ull) {   // nothing to evaluate in this chunk: only keep the window moving (cheap test below)
                if (__builtin_amdgcn_readlane(wrow, 63) >= g0 + 64) continue;
            }
            // values of this chunk's non-zeros -> slot[row - g0]; advance the window while it ends inside the chunk
            sl[lane] = 0.0;
            __builtin_amdgcn_wave_barrier();
            while (true) {
                if (wrow >= g0 && wrow < g0 + 64) sl[wrow - g0] = wval;
                const int last = __builtin_amdgcn_readlane(wrow, 63);
                if (last >= g0 + 64 || qbase + 64 >= qend) break;   // the window reaches past the chunk, or the column is exhausted
                qbase += 64;
                wrow = (qbase + lane < qend) ? Ai[qbase + lane] : INT32_MAX;
                wval = (qbase + lane < qend) ? Ax[qbase + lane] : 0.0;
            }
            __builtin_amdgcn_wave_barrier();
            const double myval = sl[lane];          // A[g0 + lane, cell] (0 if absent)
            __builtin_amdgcn_wave_barrier();
            while (mk != 0ull) {
                const int b = __builtin_ctzll(mk);
                mk &= mk - 1ull;
                const double* wd = Wd + (int64_t)(g0 + b) * k;
                double prod = 0.0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int f = lane + 64 * r;
                    if (f < k) prod = fma(wd[f], h[r], prod);
                }
                const double pred = wave_sum_f64(prod);
                const int vl = __builtin_amdgcn_readlane(__double2loint(myval), b), vh = __builtin_amdgcn_readlane(__double2hiint(myval), b);
                const double e = pred - __hiloint2double(vh, vl);
                s = fma(e, e, s);
                ++cnt;
            }
        }
        if (lane == 0) losses[cell] = (cnt > 0) ? s / (double)cnt : 0.0;
    }
}

// mse_test from the mask lists (the cell-side lists of the fit, mask_gram_list.inc): one wave per cell, FOUR drawn
// genes per step -- one per 16-lane row, lane l of a row on the factors l, l + 16, ... -- so a step costs one index load,
// NJ = ceil(k / 16) loads and FMAs per lane and one 4-step row reduction for four predictions, where the hashing kernel
// above spends a whole wave (hash, 64-lane reduction) per drawn gene.  The matrix value at (gene, cell) comes from the
// same sliding 64-entry window over the cell's ascending non-zeros (the listed genes ascend too).  The squared errors
// of a cell are summed in four interleaved partial sums (rounding only).
template <int NJ>
__global__ __launch_bounds__(256) void mse_test_list_kernel(const double* __restrict__ Ax, const int32_t* __restrict__ Ai,
                                                            const int64_t* __restrict__ Ap, int64_t n,
                                                            const int64_t* __restrict__ lptr, const int32_t* __restrict__ lidx,
                                                            const double* __restrict__ Wd, const double* __restrict__ H, int k,
                                                            double* __restrict__ losses) {
    const int lane = threadIdx.x & 63, rowid = lane >> 4, l16 = lane & 15;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t cell = gw; cell < n; cell += nwaves) {
        const int64_t qend = Ap[cell + 1];
        int64_t qbase = Ap[cell];
        const int64_t p0 = lptr[cell];
        const int nl = (int)(lptr[cell + 1] - p0);
        double h[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) h[j] = (l16 + 16 * j < k) ? H[cell * k + l16 + 16 * j] : 0.0;
        int wrow = (qbase + lane < qend) ? Ai[qbase + lane] : INT32_MAX;
        double wval = (qbase + lane < qend) ? Ax[qbase + lane] : 0.0;
        double s = 0.0;
        for (int t = 0; t < nl; t += 4) {
            const bool valid = t + rowid < nl;
            const int g = valid ? lidx[p0 + t + rowid] : 0;
            const double* wd = Wd + (int64_t)g * k + l16;
            double prod = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (l16 + 16 * j < k) prod = fma(wd[16 * j], h[j], prod);
            prod += dpp_mov_f64(prod, 0, 0xf);
            prod += dpp_mov_f64(prod, 1, 0xf);
            prod += dpp_mov_f64(prod, 2, 0xf);
            prod += dpp_mov_f64(prod, 3, 0xf);   // every lane of a 16-lane row holds its gene's prediction
            double val = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (t + r < nl) {   // (wave-uniform)
                    const int gr = __builtin_amdgcn_readlane(g, 16 * r);
                    while (__builtin_amdgcn_readlane(wrow, 63) < gr && qbase + 64 < qend) {   // slide the window up to the gene
                        qbase += 64;
                        wrow = (qbase + lane < qend) ? Ai[qbase + lane] : INT32_MAX;
                        wval = (qbase + lane < qend) ? Ax[qbase + lane] : 0.0;
                    }
                    const unsigned long long m = __ballot(wrow == gr);
                    double vr = 0.0;
                    if (m != 0ull) {
                        const int src = __builtin_ctzll(m);
                        vr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wval), src), __builtin_amdgcn_readlane(__double2loint(wval), src));
                    }
                    val = (rowid == r) ? vr : val;
                }
            }
            const double e = prod - val;
            s = valid ? fma(e, e, s) : s;
        }
        // the four rows' partial sums in row order
        const double s0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 0), __builtin_amdgcn_readlane(__double2loint(s), 0));
        const double s1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 16), __builtin_amdgcn_readlane(__double2loint(s), 16));
        const double s2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 32), __builtin_amdgcn_readlane(__double2loint(s), 32));
        const double s3 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 48), __builtin_amdgcn_readlane(__double2loint(s), 48));
        if (lane == 0) losses[cell] = (nl > 0) ? (((s0 + s1) + s2) + s3) / (double)nl : 0.0;
    }
}

// The matrix value at every listed (gene, cell) of the cell-side lists -- 0 where the cell has no non-zero at the gene --
// found ONCE per mask: the test error is traced several times per fit and the mask comes back fit after fit
// (sgl_mask_list_select), and finding A(gene, cell) is most of what a trace costs in mse_test_list_kernel (per listed
// gene a readlane, a ballot and two more readlanes, serial in the four genes of a step).
__global__ __launch_bounds__(256) void mask_vals_kernel(const double* __restrict__ Ax, const int32_t* __restrict__ Ai,
                                                        const int64_t* __restrict__ Ap, int64_t n,
                                                        const int64_t* __restrict__ lptr, const int32_t* __restrict__ lidx,
                                                        double* __restrict__ lval) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t cell = gw; cell < n; cell += nwaves) {
        const int64_t q0 = Ap[cell], qend = Ap[cell + 1];
        const int64_t p0 = lptr[cell];
        const int nl = (int)(lptr[cell + 1] - p0);
        // both ascend: lane t of a step looks its gene up by bisection in the cell's non-zeros (<= 11 + 1 probes of L2)
        for (int t = lane; t < nl; t += 64) {
            const int g = lidx[p0 + t];
            int64_t lo = q0, hi = qend;   // first non-zero with row >= g
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (Ai[mid] < g) lo = mid + 1; else hi = mid;
            }
            lval[p0 + t] = (lo < qend && Ai[lo] == g) ? Ax[lo] : 0.0;
        }
    }
}

// mse_test_list_kernel with the matrix values listed (mask_vals_kernel): same predictions, same four partial sums in the
// same order -- the same bits.
template <int NJ>
__global__ __launch_bounds__(256) void mse_test_vals_kernel(int64_t n, const int64_t* __restrict__ lptr, const int32_t* __restrict__ lidx,
                                                            const double* __restrict__ lval, const double* __restrict__ Wd,
                                                            const double* __restrict__ H, int k, double* __restrict__ losses) {
    const int lane = threadIdx.x & 63, rowid = lane >> 4, l16 = lane & 15;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t cell = gw; cell < n; cell += nwaves) {
        const int64_t p0 = lptr[cell];
        const int nl = (int)(lptr[cell + 1] - p0);
        double h[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) h[j] = (l16 + 16 * j < k) ? H[cell * k + l16 + 16 * j] : 0.0;
        double s = 0.0;
        for (int t = 0; t < nl; t += 4) {
            const bool valid = t + rowid < nl;
            const int g = valid ? lidx[p0 + t + rowid] : 0;
            const double val = valid ? lval[p0 + t + rowid] : 0.0;
            const double* wd = Wd + (int64_t)g * k + l16;
            double prod = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (l16 + 16 * j < k) prod = fma(wd[16 * j], h[j], prod);
            prod += dpp_mov_f64(prod, 0, 0xf);
            prod += dpp_mov_f64(prod, 1, 0xf);
            prod += dpp_mov_f64(prod, 2, 0xf);
            prod += dpp_mov_f64(prod, 3, 0xf);   // every lane of a 16-lane row holds its gene's prediction
            const double e = prod - val;
            s = valid ? fma(e, e, s) : s;
        }
        const double s0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 0), __builtin_amdgcn_readlane(__double2loint(s), 0));
        const double s1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 16), __builtin_amdgcn_readlane(__double2loint(s), 16));
        const double s2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 32), __builtin_amdgcn_readlane(__double2loint(s), 32));
        const double s3 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(s), 48), __builtin_amdgcn_readlane(__double2loint(s), 48));
        if (lane == 0) losses[cell] = (nl > 0) ? (((s0 + s1) + s2) + s3) / (double)nl : 0.0;
    }
}

__global__ __launch_bounds__(256) void sum_partial_kernel(const double* __restrict__ v, int64_t n,
                                                          double* __restrict__ part) {
    double s = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) s += v[t];
    __shared__ double sm[256];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sm[0];
}

__global__ void sum_final_kernel(const double* __restrict__ part, int nblocks, double* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += part[b];
    out[0] = s;
}

// out_dev[0] = sum over local cells of the per-cell mean squared test error
// (the caller divides by the global number of cells, src/singlet.cpp:567).
int k_mse_test(sgl_ctx* c, const double* Wd, const double* H, int k, uint64_t seed, uint64_t inv_density,
               double* out_dev) {
    const int64_t n = c->A.ncol;
    if (n <= 0) return SGL_OK;
    int nblocks = (int)((n + 4095) / 4096);
    if (nblocks > 256) nblocks = 256;
    if (nblocks < 1) nblocks = 1;
    SGLCHK(sgl_ws_reserve(c, sizeof(double) * ((size_t)n + (size_t)nblocks)));
    double* losses = c->ws;
    double* part = c->ws + n;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 256 * 16) blocks = 256 * 16;
    // the cell-side mask lists of this fit, when they are there (built by the H-update of the masked iteration)
    const DevMaskList& L = c->ML[0];
    const bool lists = k <= 128 && L.mask_t == 0 && L.ptr != nullptr && L.idx != nullptr && L.seed == seed && L.inv == inv_density &&
                       L.ncol == n && L.nrow == c->A.nrow && L.col_off == c->cell_offset && L.row_off == 0 && !getenv("SGL_MSE_NO_LIST");
    // ... and the matrix values at the listed entries, found once per mask (8 B per drawn pair: 12 GB at config 5; not when
    // that is more than a fifth of the free memory, SGL_MSE_NO_VALS=1: the window kernel looks them up in every trace)
    DevMaskList& Lw = c->ML[0];
    if (lists && !Lw.val_ok && !Lw.val_refused && !getenv("SGL_MSE_NO_VALS")) {
        const size_t want = (size_t)Lw.total + 64;
        if (Lw.cap_val < want) {
            if (Lw.val) (void)sgl_pool_free(Lw.val);
            Lw.val = nullptr; Lw.cap_val = 0;
            size_t free_b = 0, total_b = 0;
            HIPCHK(sgl_pool_mem_info(&free_b, &total_b));
            const size_t cap = (size_t)((double)Lw.total * 1.02) + 1024;
            if (8.0 * (double)cap > 0.2 * (double)free_b || sgl_pool_malloc(&Lw.val, sizeof(double) * cap) != hipSuccess) {
                (void)hipGetLastError();
                Lw.val = nullptr;
                Lw.val_refused = true;
            } else {
                Lw.cap_val = cap;
            }
        }
        if (Lw.val) {
            mask_vals_kernel<<<dim3((unsigned)blocks), dim3(256), 0, c->stream>>>(c->A.x, c->A.i, c->A.p, n, Lw.ptr, Lw.idx, Lw.val);
            HIPCHK(hipGetLastError());
            Lw.val_ok = true;
        }
    }
    if (lists && Lw.val_ok) {
#define SGL_MSEV(NJ_) mse_test_vals_kernel<NJ_><<<dim3((unsigned)blocks), dim3(256), 0, c->stream>>>(n, L.ptr, L.idx, L.val, Wd, H, k, losses)
        switch ((k + 15) / 16) {
            case 1: SGL_MSEV(1); break;
            case 2: SGL_MSEV(2); break;
            case 3: SGL_MSEV(3); break;
            case 4: SGL_MSEV(4); break;
            case 5: SGL_MSEV(5); break;
            case 6: SGL_MSEV(6); break;
            case 7: SGL_MSEV(7); break;
            default: SGL_MSEV(8); break;
        }
#undef SGL_MSEV
    } else if (lists) {
#define SGL_MSEL(NJ_) mse_test_list_kernel<NJ_><<<dim3((unsigned)blocks), dim3(256), 0, c->stream>>>(c->A.x, c->A.i, c->A.p, n, L.ptr, L.idx, Wd, H, k, losses)
        switch ((k + 15) / 16) {
            case 1: SGL_MSEL(1); break;
            case 2: SGL_MSEL(2); break;
            case 3: SGL_MSEL(3); break;
            case 4: SGL_MSEL(4); break;
            case 5: SGL_MSEL(5); break;
            case 6: SGL_MSEL(6); break;
            case 7: SGL_MSEL(7); break;
            default: SGL_MSEL(8); break;
        }
#undef SGL_MSEL
    } else if (k <= 64)
        mse_test_kernel<1><<<dim3((unsigned)blocks), dim3(256), 0, c->stream>>>(c->A.x, c->A.i, c->A.p, c->A.nrow, n, c->cell_offset,
                                                                                Wd, H, k, seed, sgl_div_make(inv_density), losses);
    else if (k <= 128)
        mse_test_kernel<2><<<dim3((unsigned)blocks), dim3(256), 0, c->stream>>>(c->A.x, c->A.i, c->A.p, c->A.nrow, n, c->cell_offset,
                                                                                Wd, H, k, seed, sgl_div_make(inv_density), losses);
    else if (k <= 256)
        mse_test_kernel<4><<<dim3((unsigned)blocks), dim3(256), 0, c->stream>>>(c->A.x, c->A.i, c->A.p, c->A.nrow, n, c->cell_offset,
                                                                                Wd, H, k, seed, sgl_div_make(inv_density), losses);
    else
        mse_test_kernel<16><<<dim3((unsigned)blocks), dim3(256), 0, c->stream>>>(c->A.x, c->A.i, c->A.p, c->A.nrow, n, c->cell_offset,
                                                                                 Wd, H, k, seed, sgl_div_make(inv_density), losses);
    HIPCHK(hipGetLastError());
    sum_partial_kernel<<<dim3(nblocks), dim3(256), 0, c->stream>>>(losses, n, part);
    HIPCHK(hipGetLastError());
    sum_final_kernel<<<dim3(1), dim3(64), 0, c->stream>>>(part, nblocks, out_dev);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}
