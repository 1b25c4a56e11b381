// Input staging on the resident shard: the two value transforms the R side
// applies to the dgCMatrix before run_nmf --
//   * Seurat::LogNormalize as PreprocessData.dgCMatrix calls it (R/PreprocessData.R:34-39):
//       x <- log1p(x / colSums(A)[cell] * scale_factor)
//   * weight_by_split (src/singlet.cpp:119-144): every group of cells is rescaled so that its total
//       equals that of group 0:  x <- x / (sum_g / sum_0)  for cells of group g != 0.
// Both keep the sparsity structure, so A and its transpose are transformed in place (the transpose
// looks the per-cell factor up through its row index).  HBM-bound streaming work.
#include "sgl_internal.h"

// one wave per column: sums[c] = sum of the column's values (fixed tree order -> deterministic)
__global__ __launch_bounds__(256) void colsum_kernel(const double* __restrict__ x, const int64_t* __restrict__ p, int64_t ncol,
                                                     double* __restrict__ sums) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t c = wave; c < ncol; c += nwaves) {
        double s = 0.0;
        for (int64_t e = p[c] + lane; e < p[c + 1]; e += 64) s += x[e];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) sums[c] = s;
    }
}

// MODE 0: x <- log1p(x / f[cell] * scale);  MODE 1: x <- x / f[cell]   (f[cell] == 1 leaves x as is)
template <int MODE, bool BY_ROW>
__global__ __launch_bounds__(256) void cell_factor_kernel(double* __restrict__ x, const int32_t* __restrict__ idx,
                                                          const int64_t* __restrict__ p, int64_t ncol, int64_t nnz,
                                                          const double* __restrict__ f, double scale) {
    if (BY_ROW) {  // transpose: the cell is the row index of the entry
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
            const double fc = f[idx[e]];
            x[e] = (MODE == 0) ? log1p(x[e] / fc * scale) : ((fc != 1.0) ? x[e] / fc : x[e]);
        }
    } else {       // A: the cell is the column; one wave per column
        const int lane = threadIdx.x & 63;
        const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
        for (int64_t c = wave; c < ncol; c += nwaves) {
            const double fc = f[c];
            if (MODE == 1 && fc == 1.0) continue;
            for (int64_t e = p[c] + lane; e < p[c + 1]; e += 64)
                x[e] = (MODE == 0) ? log1p(x[e] / fc * scale) : x[e] / fc;
        }
    }
}

static unsigned wave_blocks(int64_t ncol) {
    int64_t b = (ncol + 3) / 4;
    if (b > 256 * 64) b = 256 * 64;
    return (unsigned)(b < 1 ? 1 : b);
}

int k_colsum(hipStream_t s, const DevCSC& M, double* sums) {
    if (M.ncol <= 0) return SGL_OK;
    colsum_kernel<<<dim3(wave_blocks(M.ncol)), dim3(256), 0, s>>>(M.x, M.p, M.ncol, sums);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// apply per-cell factors f (device, one per column of A) to A (by column) and At (by row index)
int k_cell_factor(hipStream_t s, DevCSC& A, DevCSC& At, const double* f, int mode, double scale) {
    if (A.nnz > 0) {
        if (mode == 0) cell_factor_kernel<0, false><<<dim3(wave_blocks(A.ncol)), dim3(256), 0, s>>>(A.x, A.i, A.p, A.ncol, A.nnz, f, scale);
        else cell_factor_kernel<1, false><<<dim3(wave_blocks(A.ncol)), dim3(256), 0, s>>>(A.x, A.i, A.p, A.ncol, A.nnz, f, scale);
        HIPCHK(hipGetLastError());
    }
    if (At.nnz > 0 && At.x != nullptr) {
        int64_t blocks = (At.nnz + 255) / 256;
        if (blocks > 256 * 64) blocks = 256 * 64;
        if (mode == 0) cell_factor_kernel<0, true><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(At.x, At.i, At.p, At.ncol, At.nnz, f, scale);
        else cell_factor_kernel<1, true><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(At.x, At.i, At.p, At.ncol, At.nnz, f, scale);
        HIPCHK(hipGetLastError());
    }
    return SGL_OK;
}

// predict_link (src/singlet.cpp:429-430): the first link_rows entries of every right-hand side are
// multiplied by the matching column of the link matrix (link_rows x ncols, column-major)
__global__ void link_mul_kernel(double* __restrict__ B, const double* __restrict__ L, int k, int link_rows, int64_t ncols) {
    const int64_t total = ncols * link_rows;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = e / link_rows;
        const int j = (int)(e - c * link_rows);
        B[c * k + j] *= L[e];
    }
}

int k_link_mul(hipStream_t s, double* B, const double* L, int k, int link_rows, int64_t ncols) {
    if (ncols <= 0 || link_rows <= 0) return SGL_OK;
    int64_t blocks = (ncols * link_rows + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    link_mul_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(B, L, k, link_rows, ncols);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}
