// Dense FP64 helpers of the ALS loop: AAt (src/singlet.cpp:200-206), scale
// (:219-225), cor (:184-197).  All reductions are two-stage with a fixed
// order, so results are run-to-run deterministic.
#include "sgl_internal.h"

// ---------------------------------------------------------------- Gram ------
// G = F F^T for F k x cols (column-major).  The one dense contraction of the
// c_nmf path: done on the FP64 matrix cores, v_mfma_f64_16x16x4_f64.
//   D(16x16) += A(16x4) * B(4x16); per lane one f64 of A and one of B:
//   A[row = lane & 15][kk = lane >> 4], B[kk = lane >> 4][col = lane & 15];
//   D: 4 f64 per lane, D[row = (lane >> 4) + 4 * r][col = lane & 15]
//   (cdna_hip_programming.md:247-249).
// With A = F[16 rows of block bi, 4 columns c..c+3] and B = the same columns of
// row block bj transposed, both operands are ONE load of F each:
//   a = F[bi*16 + (lane&15), c + (lane>>4)],  b = F[bj*16 + (lane&15), c + (lane>>4)].
// A workgroup of 4 waves walks a contiguous chunk of columns; wave w takes
// columns c0 + 4*w, + 16, ...; every wave keeps all NT*(NT+1)/2 lower-triangle
// 16x16 tiles in registers (NT = ceil(k/16) <= 4 -> at most 10 tiles = 80 VGPRs).
// Partials go to ws[block][k*k] and are summed in block order by gram_reduce.
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const double* __restrict__ F, int k, int64_t cols,
                                                        int64_t cols_per_block, double* __restrict__ part) {
    constexpr int NTILES = NT * (NT + 1) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kk = lane >> 4;
    const int64_t c_begin = (int64_t)blockIdx.x * cols_per_block;
    int64_t c_end = c_begin + cols_per_block;
    if (c_end > cols) c_end = cols;

    d4 acc[NTILES];
#pragma unroll
    for (int t = 0; t < NTILES; ++t) acc[t] = d4{0, 0, 0, 0};

    for (int64_t c = c_begin + 4 * wave; c < c_end; c += 16) {
        const int64_t cc = c + kk;
        double f[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int row = b * 16 + r16;
            f[b] = (cc < c_end && row < k) ? F[cc * k + row] : 0.0;
        }
        int t = 0;
#pragma unroll
        for (int bi = 0; bi < NT; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) {
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[bi], f[bj], acc[t], 0, 0, 0);
                ++t;
            }
    }

    // reduce the 4 waves through LDS, then write the block's partial (full k x k, both triangles)
    __shared__ double sm[4][64 * 4];
    double* out = part + (size_t)blockIdx.x * k * k;
    int t = 0;
#pragma unroll
    for (int bi = 0; bi < NT; ++bi)
#pragma unroll
        for (int bj = 0; bj <= bi; ++bj) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) sm[wave][lane * 4 + r] = acc[t][r];
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = ((sm[0][lane * 4 + r] + sm[1][lane * 4 + r]) + sm[2][lane * 4 + r]) + sm[3][lane * 4 + r];
                    const int row = bi * 16 + kk + 4 * r;  // D row (lane>>4) + 4r
                    const int col = bj * 16 + r16;         // D col lane & 15
                    if (row < k && col < k) {
                        out[(size_t)col * k + row] = v;
                        if (bi != bj) out[(size_t)row * k + col] = v;
                    }
                }
            }
            ++t;
        }
}

// 64 < k <= 128 (NT = 5 .. 8): all NT (NT + 1) / 2 tiles no longer fit one wave's registers, so the tiles
// are dealt out to the 4 waves of the workgroup (tile t to wave t % 4) and every wave walks ALL columns of
// the block's chunk (the operand loads of the 4 waves hit the same cache lines).  No cross-wave reduction:
// each tile of the block's partial has one owner.
template <int NT, int W>
__device__ __forceinline__ void gram_split_wave(const double* __restrict__ F, int k, int64_t c_begin, int64_t c_end,
                                                double* __restrict__ out) {
    constexpr int NTILES = NT * (NT + 1) / 2;
    constexpr int MINE = (NTILES - W + 3) / 4;  // tiles t = W, W + 4, ...
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, kk = lane >> 4;
    d4 acc[MINE];
#pragma unroll
    for (int q = 0; q < MINE; ++q) acc[q] = d4{0, 0, 0, 0};
    for (int64_t c = c_begin; c < c_end; c += 4) {
        const int64_t cc = c + kk;
        double f[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int row = b * 16 + r16;
            f[b] = (cc < c_end && row < k) ? F[cc * k + row] : 0.0;
        }
        int t = 0, q = 0;
#pragma unroll
        for (int bi = 0; bi < NT; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) {
                if (t % 4 == W) {
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[bi], f[bj], acc[q], 0, 0, 0);
                    ++q;
                }
                ++t;
            }
    }
    int t = 0, q = 0;
#pragma unroll
    for (int bi = 0; bi < NT; ++bi)
#pragma unroll
        for (int bj = 0; bj <= bi; ++bj) {
            if (t % 4 == W) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = bi * 16 + kk + 4 * r;  // D row (lane >> 4) + 4 r
                    const int col = bj * 16 + r16;         // D col lane & 15
                    if (row < k && col < k) {
                        out[(size_t)col * k + row] = acc[q][r];
                        if (bi != bj) out[(size_t)row * k + col] = acc[q][r];
                    }
                }
                ++q;
            }
            ++t;
        }
}

template <int NT>
__global__ __launch_bounds__(256) void gram_mfma_split_kernel(const double* __restrict__ F, int k, int64_t cols,
                                                              int64_t cols_per_block, double* __restrict__ part) {
    const int64_t c_begin = (int64_t)blockIdx.x * cols_per_block;
    int64_t c_end = c_begin + cols_per_block;
    if (c_end > cols) c_end = cols;
    double* out = part + (size_t)blockIdx.x * k * k;
    switch (threadIdx.x >> 6) {
        case 0: gram_split_wave<NT, 0>(F, k, c_begin, c_end, out); break;
        case 1: gram_split_wave<NT, 1>(F, k, c_begin, c_end, out); break;
        case 2: gram_split_wave<NT, 2>(F, k, c_begin, c_end, out); break;
        default: gram_split_wave<NT, 3>(F, k, c_begin, c_end, out); break;
    }
}

// generic VALU fallback for k > 128: each launch forms the entries [pair0, pair0 + 256 * GV_MAXP) of the k x k
// product (a rank above 256 takes several launches, each reading F again: the slow, any-rank path).
#define GV_MAXP 256
__global__ __launch_bounds__(256) void gram_valu_kernel(const double* __restrict__ F, int k, int64_t cols,
                                                        int64_t cols_per_block, double* __restrict__ part, int pair0) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* tile = reinterpret_cast<double*>(smem_raw);  // [TC][k]
    constexpr int TC = 16;
    const int64_t c_begin = (int64_t)blockIdx.x * cols_per_block;
    int64_t c_end = c_begin + cols_per_block;
    if (c_end > cols) c_end = cols;
    const int npairs = k * k;
    const int pend = (pair0 + 256 * GV_MAXP < npairs) ? pair0 + 256 * GV_MAXP : npairs;
    double acc[GV_MAXP];
#pragma unroll 1
    for (int q = 0; q < GV_MAXP; ++q) acc[q] = 0.0;
    for (int64_t c0 = c_begin; c0 < c_end; c0 += TC) {
        const int nc = (int)((c_end - c0 < TC) ? (c_end - c0) : TC);
        __syncthreads();
        for (int e = threadIdx.x; e < nc * k; e += 256) tile[e] = F[c0 * k + e];
        __syncthreads();
        int q = 0;
        for (int pr = pair0 + (int)threadIdx.x; pr < pend; pr += 256, ++q) {
            const int i = pr % k, j = pr / k;
            double a = acc[q];
            for (int cc = 0; cc < nc; ++cc) a = fma(tile[cc * k + i], tile[cc * k + j], a);
            acc[q] = a;
        }
    }
    double* out = part + (size_t)blockIdx.x * k * k;
    int q = 0;
    for (int pr = pair0 + (int)threadIdx.x; pr < pend; pr += 256, ++q) out[pr] = acc[q];
}

// Fixed-order sum of per-block partials in two stages (one thread walking 1024 partials of an entry took 0.24 ms per
// Gram at config 3 -- more than the MFMA kernel that made them):  stage 1, blockIdx.y = segment s of SEG consecutive
// blocks: tmp[s][e] = sum of its blocks in order;  stage 2: out[e] = sum over the segments in order (+ the ridge on
// the diagonal of a k x k result when diag_k > 0).
#define SGL_RED_SEG 32
__global__ void partial_sum_stage1_kernel(const double* __restrict__ part, int nblocks, int n, double* __restrict__ tmp) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int b0 = blockIdx.y * SGL_RED_SEG;
    const int b1 = (b0 + SGL_RED_SEG < nblocks) ? b0 + SGL_RED_SEG : nblocks;
    double s = 0.0;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {   // eight loads in flight, added in block order
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u) * n + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < b1; ++b) s += part[(size_t)b * n + e];
    tmp[(size_t)blockIdx.y * n + e] = s;
}
__global__ void partial_sum_stage2_kernel(const double* __restrict__ tmp, int nseg, int n, int diag_k, double diag_add,
                                          double* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    double s = 0.0;
    for (int q = 0; q < nseg; ++q) s += tmp[(size_t)q * n + e];
    if (diag_k > 0 && e % diag_k == e / diag_k) s += diag_add;
    out[e] = s;
}
// Short vectors (the k row sums: n x segments <= 1024) in ONE launch, the same two stages inside one workgroup -- thread
// (segment s, entry e) sums its segment's blocks in order, then thread e the segments in order: the bits of the two-kernel
// form, one launch instead of two (round 5: at the 125 000-cell shard of an 8-GPU team the small launches of an iteration add up
// to a third of a millisecond).  add_all: + 1e-15 on every entry (scale's d, src/singlet.cpp:221) instead of a separate kernel.
__global__ __launch_bounds__(1024) void partial_sum_small_kernel(const double* __restrict__ part, int nblocks, int n, int nseg, double add_all,
                                                                 double* __restrict__ out) {
    __shared__ double tmp[1024];
    const int t = threadIdx.x;
    const int sgm = t / n, e = t - sgm * n;
    if (sgm < nseg) {
        const int b0 = sgm * SGL_RED_SEG;
        const int b1 = (b0 + SGL_RED_SEG < nblocks) ? b0 + SGL_RED_SEG : nblocks;
        double s = 0.0;
        int b = b0;
        for (; b + 8 <= b1; b += 8) {   // eight loads in flight, added in block order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u) * n + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < b1; ++b) s += part[(size_t)b * n + e];
        tmp[t] = s;
    }
    __syncthreads();
    if (t < n) {
        double s = 0.0;
        for (int q = 0; q < nseg; ++q) s += tmp[q * n + t];
        out[t] = s + add_all;
    }
}
// part: nblocks x n partials at the start of c->ws; the segment sums go behind them (the caller reserved
// sgl_partial_ws(nblocks, n) doubles)
static size_t sgl_partial_ws(int nblocks, int n) { return (size_t)nblocks * n + (size_t)((nblocks + SGL_RED_SEG - 1) / SGL_RED_SEG) * n; }
static int sgl_partial_sum(sgl_ctx* c, int nblocks, int n, int diag_k, double diag_add, double* out, double add_all = 0.0) {
    const int nseg = (nblocks + SGL_RED_SEG - 1) / SGL_RED_SEG;
    if (diag_k == 0 && n * nseg <= 1024) {
        partial_sum_small_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(c->ws, nblocks, n, nseg, add_all, out);
        HIPCHK(hipGetLastError());
        return SGL_OK;
    }
    if (add_all != 0.0) { sgl_set_error("partial sum: add_all only in the short form"); return SGL_EINVAL; }
    double* tmp = c->ws + (size_t)nblocks * n;
    partial_sum_stage1_kernel<<<dim3((n + 255) / 256, nseg), dim3(256), 0, c->stream>>>(c->ws, nblocks, n, tmp);
    HIPCHK(hipGetLastError());
    partial_sum_stage2_kernel<<<dim3((n + 255) / 256), dim3(256), 0, c->stream>>>(tmp, nseg, n, diag_k, diag_add, out);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

int k_gram_big_partials(hipStream_t s, const double* F, int k, int64_t cols, int64_t cols_per_block, int nblocks, double* part);   // kernels_gram_big.hip

int k_gram(sgl_ctx* c, const double* F, int k, int64_t cols, double* G, double diag_add) {
    if (k <= 0 || k > SGL_MAX_K) { sgl_set_error("k_gram: k=%d out of range", k); return SGL_EINVAL; }
    int nblocks = (int)((cols + 255) / 256);        // (at least 16 columns per block; fills the chip from ~65 000 columns on)
    if (nblocks > 1024) nblocks = 1024;
    const bool big_mfma = k > 128 && k <= 256 && !getenv("SGL_GRAM_BIG_VALU");   // ranks 129 - 256 on the matrix cores (kernels_gram_big.hip)
    if (k > 128 && nblocks > (big_mfma ? 512 : 128)) nblocks = big_mfma ? 512 : 128;   // k x k doubles of workspace per block
    if (nblocks < 1) nblocks = 1;
    int64_t cpb = (cols + nblocks - 1) / nblocks;
    cpb = (cpb + 15) / 16 * 16;  // whole 16-column steps per block
    nblocks = (int)((cols + cpb - 1) / cpb);
    if (nblocks < 1) nblocks = 1;
    SGLCHK(sgl_ws_reserve(c, sizeof(double) * sgl_partial_ws(nblocks, k * k)));
    hipStream_t s = c->stream;
    const int NT = (k + 15) / 16;
    if (NT == 1) gram_mfma_kernel<1><<<dim3(nblocks), dim3(256), 0, s>>>(F, k, cols, cpb, c->ws);
    else if (NT == 2) gram_mfma_kernel<2><<<dim3(nblocks), dim3(256), 0, s>>>(F, k, cols, cpb, c->ws);
    else if (NT == 3) gram_mfma_kernel<3><<<dim3(nblocks), dim3(256), 0, s>>>(F, k, cols, cpb, c->ws);
    else if (NT == 4) gram_mfma_kernel<4><<<dim3(nblocks), dim3(256), 0, s>>>(F, k, cols, cpb, c->ws);
    else if (NT == 5) gram_mfma_split_kernel<5><<<dim3(nblocks), dim3(256), 0, s>>>(F, k, cols, cpb, c->ws);
    else if (NT == 6) gram_mfma_split_kernel<6><<<dim3(nblocks), dim3(256), 0, s>>>(F, k, cols, cpb, c->ws);
    else if (NT == 7) gram_mfma_split_kernel<7><<<dim3(nblocks), dim3(256), 0, s>>>(F, k, cols, cpb, c->ws);
    else if (NT == 8) gram_mfma_split_kernel<8><<<dim3(nblocks), dim3(256), 0, s>>>(F, k, cols, cpb, c->ws);
    else if (big_mfma) SGLCHK(k_gram_big_partials(s, F, k, cols, cpb, nblocks, c->ws));
    else {
        const size_t lds = sizeof(double) * 16 * (size_t)k;
        if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_valu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        for (int pair0 = 0; pair0 < k * k; pair0 += 256 * GV_MAXP)
            gram_valu_kernel<<<dim3(nblocks), dim3(256), lds, s>>>(F, k, cols, cpb, c->ws, pair0);
    }
    HIPCHK(hipGetLastError());
    return sgl_partial_sum(c, nblocks, k * k, k, diag_add, G);
}

__global__ void add_diag_kernel(double* __restrict__ G, int k, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) G[(size_t)i * k + i] += v;
}

int k_gram_add_diag(hipStream_t s, double* G, int k, double v) {
    add_diag_kernel<<<dim3((k + 63) / 64), dim3(64), 0, s>>>(G, k, v);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// ---------------------------------------------------------------- row sums --
// part[block][row] = sum over the block's columns of F[row, c].  Threads:
// tx = row (strided by 64), ty = 4 column lanes.
__global__ __launch_bounds__(256) void rowsum_kernel(const double* __restrict__ F, int k, int64_t cols,
                                                     int64_t cols_per_block, double* __restrict__ part) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c_begin = (int64_t)blockIdx.x * cols_per_block;
    int64_t c_end = c_begin + cols_per_block;
    if (c_end > cols) c_end = cols;
    __shared__ double sm[4][64];
    for (int r0 = 0; r0 < k; r0 += 64) {
        const int row = r0 + tx;
        double s0 = 0.0, s1 = 0.0;
        if (row < k) {
            int64_t cc = c_begin + ty;
            for (; cc + 28 < c_end; cc += 32) {   // eight loads in flight; the two sums take them in the order of the plain loop below
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = F[(cc + 4 * u) * k + row];
#pragma unroll
                for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
            }
            for (; cc + 4 < c_end; cc += 8) {
                s0 += F[cc * k + row];
                s1 += F[(cc + 4) * k + row];
            }
            if (cc < c_end) s0 += F[cc * k + row];
        }
        __syncthreads();
        sm[ty][tx] = s0 + s1;
        __syncthreads();
        if (ty == 0 && row < k) part[(size_t)blockIdx.x * k + row] = ((sm[0][tx] + sm[1][tx]) + sm[2][tx]) + sm[3][tx];
    }
}


int k_rowsum(sgl_ctx* c, const double* F, int k, int64_t cols, double* d_out, int add_eps) {
    // enough blocks to fill the chip at shard sizes too (125 000 cells used to get 62: 72 us for a 50 MB read)
    int nblocks = (int)((cols + 511) / 512);
    if (nblocks > 512) nblocks = 512;
    if (nblocks < 1) nblocks = 1;
    const int64_t cpb = (cols + nblocks - 1) / nblocks;
    SGLCHK(sgl_ws_reserve(c, sizeof(double) * sgl_partial_ws(nblocks, k)));
    rowsum_kernel<<<dim3(nblocks), dim3(256), 0, c->stream>>>(F, k, cols, cpb, c->ws);
    HIPCHK(hipGetLastError());
    // add_eps: d += 1e-15 (src/singlet.cpp:221) inside the final stage -- short form only (k x segments <= 1024), else the caller's
    // k_scale_apply(add_eps = 1) does it as before
    const int nseg = (nblocks + SGL_RED_SEG - 1) / SGL_RED_SEG;
    if (add_eps && k * nseg > 1024) { sgl_set_error("k_rowsum: add_eps needs the short form"); return SGL_EINVAL; }
    return sgl_partial_sum(c, nblocks, k, 0, 0.0, d_out, add_eps ? 1e-15 : 0.0);
}
// whether k_rowsum(..., add_eps = 1) is available at this rank (the caller then skips the add_eps of k_scale_apply)
bool k_rowsum_can_add_eps(int k, int64_t cols) {
    int nblocks = (int)((cols + 511) / 512);
    if (nblocks > 512) nblocks = 512;
    if (nblocks < 1) nblocks = 1;
    return k * ((nblocks + SGL_RED_SEG - 1) / SGL_RED_SEG) <= 1024;
}

// d[i] += 1e-15 (once, by the add_eps kernel) and F[i, c] /= d[i]  (src/singlet.cpp:221-224)
__global__ void add_eps_kernel(double* __restrict__ d, int k) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) d[i] += 1e-15;
}

__global__ void scale_kernel(double* __restrict__ F, int k, int64_t n, const double* __restrict__ d) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        F[t] = F[t] / d[t % k];
}

int k_scale_apply(hipStream_t s, double* F, int k, int64_t cols, double* d, int add_eps) {
    if (add_eps) {
        add_eps_kernel<<<dim3((k + 63) / 64), dim3(64), 0, s>>>(d, k);
        HIPCHK(hipGetLastError());
    }
    const int64_t n = (int64_t)k * cols;
    if (n <= 0) return SGL_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    scale_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(F, k, n, d);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// ---------------------------------------------------------------- cor -------
__global__ __launch_bounds__(256) void cor_partial_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                          int64_t n, double* __restrict__ part) {
    double s[5] = {0, 0, 0, 0, 0};
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const double a = x[t], b = y[t];
        s[0] += a;
        s[1] += b;
        s[2] = fma(a, b, s[2]);
        s[3] = fma(a, a, s[3]);
        s[4] = fma(b, b, s[4]);
    }
    __shared__ double sm[5][256];
    for (int q = 0; q < 5; ++q) sm[q][threadIdx.x] = s[q];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int q = 0; q < 5; ++q) sm[q][threadIdx.x] += sm[q][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 5) part[(size_t)blockIdx.x * 5 + threadIdx.x] = sm[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void cor_final_kernel(const double* __restrict__ part, int nblocks, int64_t n, double* __restrict__ out) {
    if (blockIdx.x != 0) return;
    // the five sums in block order, one lane each, out of LDS: the partials come in with one round of loads by all 256 threads
    // (five lanes walking them from memory eight at a time took 14 us at 367 blocks; one thread walking all of them 64 us)
    __shared__ double pl[5 * 256];
    const int lane = threadIdx.x;
    double mine = 0.0;
    for (int b0 = 0; b0 < nblocks; b0 += 256) {
        const int nb = (nblocks - b0 < 256) ? nblocks - b0 : 256;
        __syncthreads();
        for (int e = lane; e < nb * 5; e += 256) pl[e] = part[(size_t)b0 * 5 + e];
        __syncthreads();
        if (lane < 5)
            for (int b = 0; b < nb; ++b) mine += pl[b * 5 + lane];
    }
    if (lane >= 64) return;
    double s[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q] = __shfl(mine, q, 64);
    if (lane != 0) return;
    const double nn = (double)n;
    // 1 - (n*sum_xy - sum_x*sum_y) / sqrt((n*sum_x2 - sum_x^2) * (n*sum_y2 - sum_y^2)), src/singlet.cpp:196;
    // written without contraction so the final formula rounds as the reference's does.
    const double num = __dsub_rn(__dmul_rn(nn, s[2]), __dmul_rn(s[0], s[1]));
    const double vx = __dsub_rn(__dmul_rn(nn, s[3]), __dmul_rn(s[0], s[0]));
    const double vy = __dsub_rn(__dmul_rn(nn, s[4]), __dmul_rn(s[1], s[1]));
    out[0] = 1.0 - num / sqrt(__dmul_rn(vx, vy));
}

int k_cor(sgl_ctx* c, const double* x, const double* y, int64_t n, double* out_dev) {
    int nblocks = (int)((n + 4095) / 4096);
    if (nblocks > 512) nblocks = 512;
    if (nblocks < 1) nblocks = 1;
    SGLCHK(sgl_ws_reserve(c, sizeof(double) * 5 * (size_t)nblocks));
    // (Round 5 tried the final stage inside the first kernel -- the block that finishes last sums the partials, a ticket counter
    //  behind them: 37.7 us per call at the 125 000-cell shard with the partials staged through LDS, 48.5 without, against
    //  10.9 + 14.4 us for these two launches: a back-to-back launch on one stream costs less than the fence, the ticket and a
    //  lone block's serial tail.  Taken out; the same held for the row sums.  profiles/README.md)
    cor_partial_kernel<<<dim3(nblocks), dim3(256), 0, c->stream>>>(x, y, n, c->ws);
    HIPCHK(hipGetLastError());
    cor_final_kernel<<<dim3(1), dim3(256), 0, c->stream>>>(c->ws, nblocks, n, out_dev);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// ---------------------------------------------------------------- misc ------
// Gpad[j + GS*i] = G[j + k*i] inside k x k, 0 outside, for i < KP, j < GS (GS = row stride >= KP); one more row,
// i = KP: the correctly rounded reciprocals of the diagonal, Gpad[j + GS*KP] = 1 / G[j, j] (1 outside).
__global__ void pad_gram_kernel(const double* __restrict__ G, int k, int KP, int GS, double* __restrict__ Gpad) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (KP + 1) * GS) return;
    const int j = e % GS, i = e / GS;
    if (i == KP) Gpad[e] = (j < k) ? __ddiv_rn(1.0, G[j + k * j]) : 1.0;
    else Gpad[e] = (i < k && j < k) ? G[j + k * i] : 0.0;
}

int k_pad_gram(hipStream_t s, const double* G, int k, int KP, int GS, double* Gpad) {
    pad_gram_kernel<<<dim3(((KP + 1) * GS + 255) / 256), dim3(256), 0, s>>>(G, k, KP, GS, Gpad);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// out (cols x rows, column-major) = in^T, in is rows x cols column-major.
__global__ void transpose_kernel(const double* __restrict__ in, int rows, int cols, double* __restrict__ out) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = t / rows, r = t - c * rows;  // in[r, c]
        out[r * cols + c] = in[t];
    }
}

int k_transpose_dense(hipStream_t s, const double* in, int rows, int cols, double* out) {
    transpose_kernel<<<dim3(1024), dim3(256), 0, s>>>(in, rows, cols, out);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// Wd[t, g] = W[t, g] * d[t]   (w_ of mse_test, src/singlet.cpp:539-542)
__global__ void wd_kernel(const double* __restrict__ W, const double* __restrict__ d, int k, int64_t n,
                          double* __restrict__ Wd) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        Wd[t] = W[t] * d[t % k];
}

int k_wd(hipStream_t s, const double* W, const double* d, int k, int64_t cols, double* Wd) {
    const int64_t n = (int64_t)k * cols;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    wd_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(W, d, k, n, Wd);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// G[i, j] = (G[i, j] / d_i) / d_j (+ diag_add on the diagonal): the Gram of a row-scaled factor from the
// Gram of the unscaled one (cell-sharded runs reduce unscaled partials, multi.hip)
__global__ void gram_rescale_kernel(double* __restrict__ G, int k, const double* __restrict__ d, double diag_add) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k * k) return;
    const int i = e % k, j = e / k;
    double v = (G[e] / d[i]) / d[j];
    if (i == j) v += diag_add;
    G[e] = v;
}

int k_gram_rescale(hipStream_t s, double* G, int k, const double* d, double diag_add) {
    gram_rescale_kernel<<<dim3((k * k + 255) / 256), dim3(256), 0, s>>>(G, k, d, diag_add);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

__global__ void i64_to_f64_kernel(const int64_t* __restrict__ in, double* __restrict__ out, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) out[t] = (double)in[t];
}
__global__ void f64_to_i64_kernel(const double* __restrict__ in, int64_t* __restrict__ out, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) out[t] = (int64_t)in[t];
}
int k_i64_to_f64(hipStream_t s, const int64_t* in, double* out, int64_t n) {
    if (n <= 0) return SGL_OK;
    i64_to_f64_kernel<<<dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s>>>(in, out, n);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}
int k_f64_to_i64(hipStream_t s, const double* in, int64_t* out, int64_t n) {
    if (n <= 0) return SGL_OK;
    f64_to_i64_kernel<<<dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s>>>(in, out, n);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}
