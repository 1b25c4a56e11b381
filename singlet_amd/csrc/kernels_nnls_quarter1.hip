// nnls_quarter_kernel<KQ> instances (nnls_quarter.h), part 1 of 2: split so that no translation unit of the build takes more
// than ~2.5 minutes to compile
#include "nnls_quarter.h"

int k_nnls_quarter_part1(hipStream_t s, const double* G, const double* B, double* X, const int64_t* col_nnz, int k, int64_t ncols, double L1, double L2,
                         unsigned long long* sweep_counter, const int32_t* order, uint8_t* prev_it) {
    switch ((k + 15) / 16) {
        case 9: return launch_quarter<36>(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, order, prev_it);
        case 10: return launch_quarter<40>(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, order, prev_it);
        case 11: return launch_quarter<44>(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, order, prev_it);
        case 12: return launch_quarter<48>(s, G, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, order, prev_it);
        default: sgl_set_error("k_nnls_quarter: k=%d outside this part's ranks", k); return SGL_EINVAL;
    }
}
