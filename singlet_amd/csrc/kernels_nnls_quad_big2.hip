// nnls_quad_global_kernel<NR> instances for ranks 129 - 256 (nnls_quad_global.h), part 2 of 2 (split to bound the compile time of a
// translation unit).  Until round 6 every solve above k = 128 fell to nnls_wave_kernel -- one wave per column, ~49 instructions per
// column and coordinate step against ~13 here.  Two waves per SIMD.  Same arithmetic, same order: the bits of the wave kernel.
#include "nnls_quad_global.h"

int k_nnls_quad_global_big2(hipStream_t s, const double* G, int64_t gstride, const double* B, double* X, const int64_t* col_nnz, int k,
                            int64_t ncols, double L1, double L2, unsigned long long* sweep_counter) {
    switch ((k + 15) / 16) {
        case 13: return launch_nnls_quad_global<13>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 14: return launch_nnls_quad_global<14>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 15: return launch_nnls_quad_global<15>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 16: return launch_nnls_quad_global<16>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        default: sgl_set_error("k_nnls_quad_global_big: k=%d outside this part's ranks", k); return SGL_EINVAL;
    }
}
