// nnls (src/singlet.cpp:229-250) at ranks 129 - 256: four columns per wave (nnls_quad_global.h), instances NR = 9 .. 16.
// Until round 6 every solve above k = 128 fell to nnls_wave_kernel -- one wave per column, ~49 instructions per column and
// coordinate step against ~13 here.  The reference's nnls has no rank limit and RunNMF hands ard_nmf k_max = 1e4
// (R/RunNMF.R:131).  With a SHARED Gram (gstride = 0: the plain fit) the rows come out of L2: k x k doubles = 0.5 MB at k = 256.
// Two waves per SIMD (b, x and two row buffers are 8 NR = 128 registers at NR = 16).  Same arithmetic, same order: the bits of
// the wave kernel (tests/test_gpu_ops.py test_nnls, k = 130 ... 256).
#include "nnls_quad_global.h"

int k_nnls_quad_global_big(hipStream_t s, const double* G, int64_t gstride, const double* B, double* X, const int64_t* col_nnz, int k,
                           int64_t ncols, double L1, double L2, unsigned long long* sweep_counter) {
    switch ((k + 15) / 16) {
        case 9: return launch_nnls_quad_global<9>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 10: return launch_nnls_quad_global<10>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 11: return launch_nnls_quad_global<11>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 12: return launch_nnls_quad_global<12>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 13: return launch_nnls_quad_global<13>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 14: return launch_nnls_quad_global<14>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 15: return launch_nnls_quad_global<15>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        case 16: return launch_nnls_quad_global<16>(s, G, gstride, B, X, col_nnz, k, ncols, L1, L2, sweep_counter);
        default: sgl_set_error("k_nnls_quad_global_big: k=%d outside 129 .. 256", k); return SGL_EINVAL;
    }
}
