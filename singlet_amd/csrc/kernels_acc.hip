// Sparse-dense accumulate of predict / predict_mask (src/singlet.cpp:341-343,
// 449-457):   B[:, c] = sum over the non-zeros (r, v) of column c, in stored
// order, of v * F[:, r].
//
// HBM-bound index/value streaming + a gather of k-vectors of the dense factor.
// Mapping: one wave per column, lanes over the k factor rows (coalesced
// k*8-byte reads of F[:, r]); the column's (row, value) pairs are read 64 at a
// time, one pair per lane (coalesced), and broadcast with v_readlane.  The row
// range is cut into tiles whose slice of F fits an XCD's L2 (DevCSC::seg);
// tiles are separate launches so that all resident waves gather from the same
// L2-resident slice of F, and the per-column sum still runs in stored order.
#include "sgl_internal.h"

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffull), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// MASK: 0 none; 1 skip entries with draw(col_global, row_global) [A pass, mask_t = false];
//       2 skip entries with draw(row_global, col_global) [At pass, mask_t = true]
template <int R, int MASK>
__global__ __launch_bounds__(256) void acc_kernel(const double* __restrict__ x, const int32_t* __restrict__ idx,
                                                  const int64_t* __restrict__ seg_lo, const int64_t* __restrict__ seg_hi,
                                                  int64_t ncols, const double* __restrict__ F, int k,
                                                  double* __restrict__ B, int accumulate, uint64_t seed,
                                                  SglDiv inv_density, int64_t col_off, int64_t row_off) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t col = wave; col < ncols; col += nwaves) {
        const int64_t lo = seg_lo[col], hi = seg_hi[col];
        if (lo == hi && accumulate) continue;
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
        if (accumulate) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int f = lane + 64 * r;
                if (f < k) acc[r] = B[col * k + f];
            }
        }
        for (int64_t base = lo; base < hi; base += 64) {
            const int64_t q = base + lane;
            int my_r = 0;
            double my_v = 0.0;
            if (q < hi) {
                my_r = idx[q];
                my_v = x[q];
                if (MASK == 1) {
                    if (sgl_draw(seed, (uint64_t)(col + col_off), (uint64_t)(my_r + row_off), inv_density)) my_v = 0.0, my_r = -1;
                } else if (MASK == 2) {
                    if (sgl_draw(seed, (uint64_t)(my_r + row_off), (uint64_t)(col + col_off), inv_density)) my_v = 0.0, my_r = -1;
                }
            }
            const int cnt = (int)((hi - base < 64) ? (hi - base) : 64);
            int t = 0;
            for (; t + 4 <= cnt; t += 4) {
                const int r0 = __builtin_amdgcn_readlane(my_r, t), r1 = __builtin_amdgcn_readlane(my_r, t + 1);
                const int r2 = __builtin_amdgcn_readlane(my_r, t + 2), r3 = __builtin_amdgcn_readlane(my_r, t + 3);
                const double v0 = readlane_f64(my_v, t), v1 = readlane_f64(my_v, t + 1);
                const double v2 = readlane_f64(my_v, t + 2), v3 = readlane_f64(my_v, t + 3);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int f = lane + 64 * r;
                    if (f < k) {
                        // masked entries (row -1) contribute nothing and must not be read
                        const double f0 = (MASK && r0 < 0) ? 0.0 : F[(int64_t)r0 * k + f];
                        const double f1 = (MASK && r1 < 0) ? 0.0 : F[(int64_t)r1 * k + f];
                        const double f2 = (MASK && r2 < 0) ? 0.0 : F[(int64_t)r2 * k + f];
                        const double f3 = (MASK && r3 < 0) ? 0.0 : F[(int64_t)r3 * k + f];
                        double a = acc[r];
                        if (!MASK || r0 >= 0) a = fma(v0, f0, a);
                        if (!MASK || r1 >= 0) a = fma(v1, f1, a);
                        if (!MASK || r2 >= 0) a = fma(v2, f2, a);
                        if (!MASK || r3 >= 0) a = fma(v3, f3, a);
                        acc[r] = a;
                    }
                }
            }
            for (; t < cnt; ++t) {
                const int r0 = __builtin_amdgcn_readlane(my_r, t);
                const double v0 = readlane_f64(my_v, t);
                if (MASK && r0 < 0) continue;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int f = lane + 64 * r;
                    if (f < k) acc[r] = fma(v0, F[(int64_t)r0 * k + f], acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int f = lane + 64 * r;
            if (f < k) B[col * k + f] = acc[r];
        }
    }
}

template <int MASK>
static int launch_acc(hipStream_t s, const DevCSC& M, int tile, const double* F, int k, double* B, uint64_t seed,
                      uint64_t inv_density, int64_t col_off, int64_t row_off) {
    const int64_t* lo = M.seg + (size_t)tile * M.ncol;
    const int64_t* hi = M.seg + (size_t)(tile + 1) * M.ncol;
    int64_t blocks = ((int64_t)M.ncol + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    const int R = (k + 63) / 64;
    const int accumulate = tile > 0;
    dim3 g((unsigned)blocks), b(256);
#define SGL_ACC(RR) acc_kernel<RR, MASK><<<g, b, 0, s>>>(M.x, M.i, lo, hi, M.ncol, F, k, B, accumulate, seed, sgl_div_make(inv_density), col_off, row_off)
    if (R == 1) SGL_ACC(1);
    else if (R == 2) SGL_ACC(2);
    else if (R == 3) SGL_ACC(3);
    else if (R == 4) SGL_ACC(4);
    else if (R <= 8) SGL_ACC(8);       // ranks above 256: the generic instances (SGL_MAX_K = 1024)
    else if (R <= 16) SGL_ACC(16);
    else { sgl_set_error("k_acc: k=%d too large", k); return SGL_EINVAL; }
#undef SGL_ACC
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

int k_acc(hipStream_t s, const DevCSC& M, const double* F, int k, double* B, uint64_t mask_seed,
          uint64_t inv_density, int mask_mode, int64_t mask_col_offset, int64_t mask_row_offset) {
    if (M.ncol <= 0) return SGL_OK;
    for (int t = 0; t < M.ntiles; ++t) {
        int rc;
        if (mask_mode == 0) rc = launch_acc<0>(s, M, t, F, k, B, 0, 1, 0, 0);
        else if (mask_mode == 1) rc = launch_acc<1>(s, M, t, F, k, B, mask_seed, inv_density, mask_col_offset, mask_row_offset);
        else rc = launch_acc<2>(s, M, t, F, k, B, mask_seed, inv_density, mask_col_offset, mask_row_offset);
        if (rc != SGL_OK) return rc;
    }
    return SGL_OK;
}
