// AAt (src/singlet.cpp:200-206) at ranks 129 - 256 on the FP64 matrix cores (round 6).  Until now a Gram above k = 128 ran on the
// VALU (gram_valu_kernel: 6.8 ms per 200 000 columns at k = 130 where k = 128 takes 0.34 on the MFMA kernel); the reference's AAt
// has no rank limit and RunNMF hands ard_nmf k_max = 1e4 (R/RunNMF.R:131).
//
// Same mapping as gram_mfma_split_kernel (kernels_dense.hip): lower-triangle 16 x 16 tiles of v_mfma_f64_16x16x4_f64, each
// with ONE owner wave, both operands one load of F each.  NT = ceil(k / 16) <= 16 gives up to 136 tiles -- 34 per wave would be
// 272 accumulator registers -- so the triangle is cut into PARTS = 4 launches' worth of tiles over blockIdx.y: tile t belongs to
// (part, wave) = ((t / 4) % 4, t % 4), at most 9 tiles = 72 registers per wave, and every part walks all columns of its chunk
// (the operand loads of the 16 owners of a chunk hit the same lines in L2).  NT is rounded up to even (four instances: 10, 12,
// 14, 16; rows past the rank load zeros).  Partials per column chunk, summed in fixed order by the caller (sgl_partial_sum).
#include "sgl_internal.h"

typedef double gb_d4 __attribute__((ext_vector_type(4)));

template <int NT, int PART, int W>
__device__ __forceinline__ void gram_big_wave(const double* __restrict__ F, int k, int64_t c_begin, int64_t c_end, double* __restrict__ out) {
    constexpr int NTILES = NT * (NT + 1) / 2;
    constexpr int OWNER = 4 * PART + W;                 // tiles t = OWNER, OWNER + 16, ...
    constexpr int MINE = (NTILES - OWNER + 15) / 16;
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, kk = lane >> 4;
    gb_d4 acc[MINE > 0 ? MINE : 1];
#pragma unroll
    for (int q = 0; q < MINE; ++q) acc[q] = gb_d4{0, 0, 0, 0};
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
                if (t % 16 == OWNER) {
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
            if (t % 16 == OWNER) {
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

template <int NT, int PART>
__device__ __forceinline__ void gram_big_part(const double* __restrict__ F, int k, int64_t c_begin, int64_t c_end, double* __restrict__ out) {
    switch (threadIdx.x >> 6) {
        case 0: gram_big_wave<NT, PART, 0>(F, k, c_begin, c_end, out); break;
        case 1: gram_big_wave<NT, PART, 1>(F, k, c_begin, c_end, out); break;
        case 2: gram_big_wave<NT, PART, 2>(F, k, c_begin, c_end, out); break;
        default: gram_big_wave<NT, PART, 3>(F, k, c_begin, c_end, out); break;
    }
}

template <int NT>
__global__ __launch_bounds__(256) void gram_mfma_big_kernel(const double* __restrict__ F, int k, int64_t cols, int64_t cols_per_block,
                                                            double* __restrict__ part) {
    const int64_t c_begin = (int64_t)blockIdx.x * cols_per_block;
    int64_t c_end = c_begin + cols_per_block;
    if (c_end > cols) c_end = cols;
    double* out = part + (size_t)blockIdx.x * k * k;
    switch (blockIdx.y) {
        case 0: gram_big_part<NT, 0>(F, k, c_begin, c_end, out); break;
        case 1: gram_big_part<NT, 1>(F, k, c_begin, c_end, out); break;
        case 2: gram_big_part<NT, 2>(F, k, c_begin, c_end, out); break;
        default: gram_big_part<NT, 3>(F, k, c_begin, c_end, out); break;
    }
}

// partials of G = F F^T for 128 < k <= 256: part[block][k * k], nblocks column chunks of cols_per_block columns
int k_gram_big_partials(hipStream_t s, const double* F, int k, int64_t cols, int64_t cols_per_block, int nblocks, double* part) {
    const dim3 g((unsigned)nblocks, 4), b(256);
    switch ((k + 31) / 32) {
        case 5: gram_mfma_big_kernel<10><<<g, b, 0, s>>>(F, k, cols, cols_per_block, part); break;
        case 6: gram_mfma_big_kernel<12><<<g, b, 0, s>>>(F, k, cols, cols_per_block, part); break;
        case 7: gram_mfma_big_kernel<14><<<g, b, 0, s>>>(F, k, cols, cols_per_block, part); break;
        case 8: gram_mfma_big_kernel<16><<<g, b, 0, s>>>(F, k, cols, cols_per_block, part); break;
        default: sgl_set_error("k_gram_big_partials: k=%d outside 129 .. 256", k); return SGL_EINVAL;
    }
    HIPCHK(hipGetLastError());
    return SGL_OK;
}
