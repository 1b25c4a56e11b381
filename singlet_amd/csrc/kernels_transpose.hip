// Device-side replacement of Matrix::t(A) (R/run_nmf.R:40): CSC of the shard
// -> CSC of its transpose, row indices ascending within each column.  A stable
// radix sort of the non-zeros by row index keeps the (ascending) column order
// inside every row, which is exactly what R's t() produces.  HBM-bound index
// work; rocPRIM (via hipcub) does the sort.
#include "sgl_internal.h"
#include <hipcub/hipcub.hpp>

__global__ void expand_cols_kernel(const int64_t* __restrict__ p, int64_t ncol, int32_t* __restrict__ colof,
                                   uint32_t* __restrict__ iota) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t c = wave; c < ncol; c += nwaves) {
        const int64_t lo = p[c], hi = p[c + 1];
        for (int64_t q = lo + lane; q < hi; q += 64) {
            colof[q] = (int32_t)c;
            iota[q] = (uint32_t)q;
        }
    }
}

__global__ void row_hist_kernel(const int32_t* __restrict__ idx, int64_t nnz, unsigned long long* __restrict__ counts) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&counts[idx[q]], 1ull);
}

__global__ void gather_kernel(const uint32_t* __restrict__ perm, int64_t nnz, const int32_t* __restrict__ colof,
                              const double* __restrict__ x, int32_t* __restrict__ ti, double* __restrict__ tx) {
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < nnz; d += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t q = perm[d];
        ti[d] = colof[q];
        tx[d] = x[q];
    }
}

template <typename T>
static int talloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = sgl_pool_malloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        sgl_set_error("hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
        return SGL_ENOMEM;
    }
    return SGL_OK;
}

// Fills c->At from c->A.
int sgl_device_transpose(sgl_ctx* c) {
    const DevCSC& A = c->A;
    DevCSC& T = c->At;
    hipStream_t s = c->stream;
    const int64_t nnz = A.nnz;
    T.nrow = A.ncol;
    T.ncol = A.nrow;
    T.nnz = nnz;
    SGLCHK(talloc(&T.x, (size_t)nnz));
    SGLCHK(talloc(&T.i, (size_t)nnz));
    SGLCHK(talloc(&T.p, (size_t)T.ncol + 1));

    int64_t* counts = nullptr;
    int32_t *colof = nullptr, *keys_out = nullptr;
    uint32_t *iota = nullptr, *perm = nullptr;
    void* tmp = nullptr;
    int rc = SGL_OK;
    do {
        if ((rc = talloc(&counts, (size_t)T.ncol)) != SGL_OK) break;
        if ((rc = talloc(&colof, (size_t)nnz)) != SGL_OK) break;
        if ((rc = talloc(&keys_out, (size_t)nnz)) != SGL_OK) break;
        if ((rc = talloc(&iota, (size_t)nnz)) != SGL_OK) break;
        if ((rc = talloc(&perm, (size_t)nnz)) != SGL_OK) break;
        if (hipMemsetAsync(counts, 0, sizeof(int64_t) * (size_t)T.ncol, s) != hipSuccess) { rc = SGL_EHIP; break; }
        if (nnz > 0) {
            int64_t blocks = std::min<int64_t>((nnz + 255) / 256, 256 * 32);
            row_hist_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(A.i, nnz, (unsigned long long*)counts);
            int64_t wb = std::min<int64_t>(((int64_t)A.ncol + 3) / 4, 256 * 32);
            expand_cols_kernel<<<dim3((unsigned)wb), dim3(256), 0, s>>>(A.p, A.ncol, colof, iota);
        }
        if ((rc = k_exclusive_scan(c, counts, T.p, T.ncol)) != SGL_OK) break;
        if ((rc = k_scan_total(s, counts, T.p, T.ncol)) != SGL_OK) break;
        if (nnz > 0) {
            int end_bit = 1;
            while (((int64_t)1 << end_bit) < (int64_t)A.nrow && end_bit < 31) ++end_bit;
            size_t tmp_bytes = 0;
            if (hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, A.i, keys_out, iota, perm, nnz, 0, end_bit, s) != hipSuccess) { rc = SGL_EHIP; break; }
            if (sgl_pool_malloc(&tmp, tmp_bytes ? tmp_bytes : 1) != hipSuccess) { (void)hipGetLastError(); sgl_set_error("transpose: temp alloc failed"); rc = SGL_ENOMEM; break; }
            if (hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, A.i, keys_out, iota, perm, nnz, 0, end_bit, s) != hipSuccess) { rc = SGL_EHIP; break; }
            int64_t blocks = std::min<int64_t>((nnz + 255) / 256, 256 * 32);
            gather_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(perm, nnz, colof, A.x, T.i, T.x);
        }
        if (hipGetLastError() != hipSuccess) { rc = SGL_EHIP; break; }
    } while (0);
    hipError_t e = hipStreamSynchronize(s);
    if (rc == SGL_EHIP || e != hipSuccess) { sgl_set_error("device transpose failed: %s", hipGetErrorString(e)); rc = SGL_EHIP; }
    if (counts) (void)sgl_pool_free(counts);
    if (colof) (void)sgl_pool_free(colof);
    if (keys_out) (void)sgl_pool_free(keys_out);
    if (iota) (void)sgl_pool_free(iota);
    if (perm) (void)sgl_pool_free(perm);
    if (tmp) (void)sgl_pool_free(tmp);
    return rc;
}
