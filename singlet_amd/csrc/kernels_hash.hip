// Integer kernels of the path: rng::rand / draw (src/singlet.cpp:30-64, 91-95),
// the hash-driven synthetic CSC generator of SURVEY.md 8(d), column-pointer
// utilities and the row-tile segment table used by the sparse accumulate.
// All bit-exact uint64 arithmetic; HBM-bound byte/index work, no MFMA.
#include "sgl_internal.h"
#include <hipcub/hipcub.hpp>

__global__ void rand_kernel(uint64_t state, const uint64_t* __restrict__ i, const uint64_t* __restrict__ j, int64_t n,
                            uint64_t* __restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = sgl_rand2(state, i[t], j[t]);
}

int k_rand(hipStream_t s, uint64_t state, const uint64_t* i, const uint64_t* j, int64_t n, uint64_t* out) {
    if (n <= 0) return SGL_OK;
    rand_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(state, i, j, n, out);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// out[c * ngenes + g] = draw(cell0 + c, g): one block row per cell, lanes over genes.
__global__ void mask_kernel(uint64_t state, SglDiv inv_density, int64_t cell0, int32_t ncells, int32_t ngenes,
                            uint8_t* __restrict__ out) {
    const int64_t c = blockIdx.y;
    const uint64_t xi = sgl_rand_i(state, (uint64_t)(cell0 + c));
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngenes; g += (int64_t)gridDim.x * blockDim.x)
        out[c * ngenes + g] = (uint8_t)sgl_divides(sgl_rand_j(xi, (uint64_t)g), inv_density);
}

int k_mask(hipStream_t s, uint64_t state, uint64_t inv_density, int64_t cell0, int32_t ncells, int32_t ngenes,
           uint8_t* out) {
    if (ncells <= 0 || ngenes <= 0) return SGL_OK;
    unsigned gx = (unsigned)((ngenes + 255) / 256);
    if (gx > 64) gx = 64;
    mask_kernel<<<dim3(gx, (unsigned)ncells), dim3(256), 0, s>>>(state, sgl_div_make(inv_density), cell0, ncells, ngenes, out);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// ---- synthetic generator ---------------------------------------------------
// One wave per column.  transposed = 0: column = cell, rows = genes;
// transposed = 1: column = gene, rows = local cells.  The hash always takes
// (global cell, gene) in that order (src/singlet.cpp:450).
//
// Skewed variant (skew != nullptr; bench.py --data skewed, the secondary record SURVEY.md 8d asks for next to the
// i.i.d. headline): entry (gene g, cell c) is non-zero iff  u(c, g) < p0 * wc[lc(c)] * wg[lg(g)]  with u the hash
// mapped to [0, 1), p0 = 1 / inv_density and wc / wg two 16-level weight tables (log-normal quantiles, mean 1)
// indexed by a hash of the cell / gene index alone: columns AND rows get heavy-tailed non-zero counts, as
// single-cell count matrices have (pbmc3k: 3 ... 2700 non-zeros per gene).  skew = [wc[16] | wg[16]] on the device.
__device__ __forceinline__ int synth_level_cell(uint64_t S, uint64_t cell) { return (int)((sgl_rand2(S + 3, cell, 0x5EEDull) >> 11) & 15); }
__device__ __forceinline__ int synth_level_gene(uint64_t S, uint64_t gene) { return (int)((sgl_rand2(S + 4, 0x5EEDull, gene) >> 11) & 15); }

template <bool FILL>
__global__ __launch_bounds__(256) void synth_kernel(uint64_t S, SglDiv inv_density, const double* __restrict__ levels,
                                                    int transposed, int64_t cell_offset, int32_t ncells, int32_t ngenes,
                                                    int64_t* __restrict__ counts, const int64_t* __restrict__ p,
                                                    int32_t* __restrict__ idx, double* __restrict__ x,
                                                    const double* __restrict__ skew) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ncol = transposed ? ngenes : ncells;
    const int64_t nrow = transposed ? ncells : ngenes;
    for (int64_t col = wave; col < ncol; col += nwaves) {
        int64_t pos = FILL ? p[col] : 0;
        int64_t cnt = 0;
        uint64_t xi = 0;
        if (!transposed) xi = sgl_rand_i(S, (uint64_t)(cell_offset + col));
        uint64_t xi1 = 0;
        if (FILL && !transposed) xi1 = sgl_rand_i(S + 1, (uint64_t)(cell_offset + col));
        double wcol = 0.0;
        const double p0 = 1.0 / (double)inv_density.d;
        if (skew) wcol = transposed ? skew[16 + synth_level_gene(S, (uint64_t)col)] : skew[synth_level_cell(S, (uint64_t)(cell_offset + col))];
        for (int64_t r0 = 0; r0 < nrow; r0 += 64) {
            const int64_t r = r0 + lane;
            bool drawn = false;
            if (r < nrow) {
                const uint64_t h = transposed ? sgl_rand2(S, (uint64_t)(cell_offset + r), (uint64_t)col)
                                              : sgl_rand_j(xi, (uint64_t)r);
                if (skew) {
                    const double wrow = transposed ? skew[synth_level_cell(S, (uint64_t)(cell_offset + r))] : skew[16 + synth_level_gene(S, (uint64_t)r)];
                    drawn = (double)(h >> 11) * 0x1p-53 < p0 * (wcol * wrow);   // wc * wg first: commutative, so both orientations round alike
                } else {
                    drawn = sgl_divides(h, inv_density);
                }
            }
            const unsigned long long m = __ballot(drawn);
            if (FILL) {
                if (drawn) {
                    const int64_t dst = pos + __popcll(m & ((1ull << lane) - 1ull));
                    const uint64_t h1 = transposed ? sgl_rand2(S + 1, (uint64_t)(cell_offset + r), (uint64_t)col)
                                                   : sgl_rand_j(xi1, (uint64_t)r);
                    idx[dst] = (int32_t)r;
                    x[dst] = levels[(h1 >> 11) & 15];
                }
                pos += __popcll(m);
            } else {
                cnt += __popcll(m);
            }
        }
        if (!FILL && lane == 0) counts[col] = cnt;
    }
}

static unsigned wave_grid(int64_t ncol) {
    int64_t blocks = (ncol + 3) / 4;  // 4 waves per 256-thread block
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

int k_synth_count(hipStream_t s, uint64_t S, uint64_t inv_density, int transposed, int64_t cell_offset,
                  int32_t ncells, int32_t ngenes, int64_t* counts, const double* skew_dev) {
    const int64_t ncol = transposed ? ngenes : ncells;
    if (ncol <= 0) return SGL_OK;
    synth_kernel<false><<<dim3(wave_grid(ncol)), dim3(256), 0, s>>>(S, sgl_div_make(inv_density), nullptr, transposed, cell_offset,
                                                                    ncells, ngenes, counts, nullptr, nullptr, nullptr, skew_dev);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

int k_synth_fill(hipStream_t s, uint64_t S, uint64_t inv_density, const double* levels16_dev, int transposed,
                 int64_t cell_offset, int32_t ncells, int32_t ngenes, const int64_t* p, int32_t* idx, double* x,
                 const double* skew_dev) {
    const int64_t ncol = transposed ? ngenes : ncells;
    if (ncol <= 0) return SGL_OK;
    synth_kernel<true><<<dim3(wave_grid(ncol)), dim3(256), 0, s>>>(S, sgl_div_make(inv_density), levels16_dev, transposed,
                                                                   cell_offset, ncells, ngenes, nullptr, p, idx, x, skew_dev);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// w[f, g] = ((rand_{S+2}(f, g) >> 11) + 0.5) * 2^-53   (stand-in for stats::runif, R/run_nmf.R:55)
__global__ void winit_kernel(uint64_t S, int k, int32_t ngenes, double* __restrict__ W) {
    const int64_t n = (int64_t)k * ngenes;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = t / k;
        const int f = (int)(t - g * k);
        W[t] = ((double)(sgl_rand2(S + 2, (uint64_t)f, (uint64_t)g) >> 11) + 0.5) * 0x1p-53;
    }
}

int k_synth_winit(hipStream_t s, uint64_t S, int k, int32_t ngenes, double* W) {
    winit_kernel<<<dim3(1024), dim3(256), 0, s>>>(S, k, ngenes, W);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// ---- column pointer utilities ---------------------------------------------
int k_exclusive_scan(sgl_ctx* c, const int64_t* in, int64_t* out, int64_t n) {
    // out[0..n]: out[t] = sum_{u<t} in[u]; out[n] = total.  hipcub scan over n+1
    // items needs in[n] readable, so scan n items and finish the total on device.
    size_t tmp = 0;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, in, out, (int)n, c->stream));
    SGLCHK(sgl_ws_reserve(c, tmp));
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(c->ws, tmp, in, out, (int)n, c->stream));
    return SGL_OK;
}

__global__ void scan_total_kernel(const int64_t* __restrict__ in, int64_t* __restrict__ out, int64_t n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (n > 0) ? out[n - 1] + in[n - 1] : 0;
}

int k_scan_total(hipStream_t s, const int64_t* in, int64_t* out, int64_t n) {
    scan_total_kernel<<<dim3(1), dim3(64), 0, s>>>(in, out, n);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

__global__ void col_counts_kernel(const int64_t* __restrict__ p, int64_t ncol, int64_t* __restrict__ counts) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ncol) counts[t] = p[t + 1] - p[t];
}

int k_col_counts(hipStream_t s, const int64_t* p, int64_t ncol, int64_t* counts) {
    if (ncol <= 0) return SGL_OK;
    col_counts_kernel<<<dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, s>>>(p, ncol, counts);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

__global__ void widen_kernel(const int32_t* __restrict__ p32, int64_t n1, int64_t* __restrict__ p64) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n1) p64[t] = (int64_t)p32[t];
}

int k_widen_p(hipStream_t s, const int32_t* p32, int64_t n1, int64_t* p64) {
    widen_kernel<<<dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s>>>(p32, n1, p64);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// seg[t * ncol + c] = first q in [p[c], p[c+1]) with idx[q] >= t * tile_rows (or p[c+1]).
__global__ void segments_kernel(const int32_t* __restrict__ idx, const int64_t* __restrict__ p, int64_t ncol,
                                int32_t tile_rows, int32_t ntiles, int64_t* __restrict__ seg) {
    const int64_t tot = (int64_t)(ntiles + 1) * ncol;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < tot; u += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = u / ncol, c = u - t * ncol;
        int64_t lo = p[c], hi = p[c + 1];
        if (t == 0) {
            seg[u] = lo;
        } else if (t == ntiles) {
            seg[u] = hi;
        } else {
            const int64_t bound = t * (int64_t)tile_rows;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)idx[mid] < bound) lo = mid + 1; else hi = mid;
            }
            seg[u] = lo;
        }
    }
}

int k_build_segments(hipStream_t s, const DevCSC& M) {
    const int64_t tot = (int64_t)(M.ntiles + 1) * M.ncol;
    if (tot <= 0) return SGL_OK;
    int64_t blocks = (tot + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    segments_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(M.i, M.p, M.ncol, M.tile_rows, M.ntiles, M.seg);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// dgCMatrix invariants the kernels rely on (a wrong row index would read or write out of bounds): row
// indices inside [0, nrow) and strictly ascending within a column.  flag |= 1 / 2.
__global__ __launch_bounds__(256) void validate_csc_kernel(const int32_t* __restrict__ idx, const int64_t* __restrict__ p,
                                                           int64_t ncol, int32_t nrow, int* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int bad = 0;
    for (int64_t c = wave; c < ncol; c += nwaves) {
        const int64_t lo = p[c], hi = p[c + 1];
        for (int64_t e = lo + lane; e < hi; e += 64) {
            const int32_t r = idx[e];
            if (r < 0 || r >= nrow) bad |= 1;
            if (e > lo && idx[e - 1] >= r) bad |= 2;
        }
    }
    if (bad) atomicOr(flag, bad);
}

// flag |= 4 when a value is NaN or +-Inf.  The reference would carry such a value into every factor (each b is a sum over
// the column, src/singlet.cpp:341-343); the shared-Gram solve here assumes finite right-hand sides (nnls_static_for.h:
// the branch-free step), so non-finite input is refused at the door instead.  Call after k_validate_csc (which clears the flag).
__global__ __launch_bounds__(256) void all_finite_kernel(const double* __restrict__ x, int64_t n, int* __restrict__ flag) {
    int bad = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[e];
        if (!(__builtin_fabs(v) < __builtin_inf())) bad = 4;
    }
    if (bad) atomicOr(flag, bad);
}

int k_all_finite(hipStream_t s, const double* x, int64_t n, int* flag_dev) {
    if (n <= 0) return SGL_OK;
    all_finite_kernel<<<dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, s>>>(x, n, flag_dev);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

int k_validate_csc(hipStream_t s, const int32_t* idx, const int64_t* p, int64_t ncol, int32_t nrow, int* flag_dev) {
    if (ncol <= 0) return SGL_OK;
    HIPCHK(hipMemsetAsync(flag_dev, 0, sizeof(int), s));
    validate_csc_kernel<<<dim3(wave_grid(ncol)), dim3(256), 0, s>>>(idx, p, ncol, nrow, flag_dev);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}
