// nnls_lane_kernel<KP> instances for KP = 2 .. 40 (see nnls_lane.h)
#include "nnls_lane.h"

int k_nnls_lane_launch1(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                        int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g,
                        dim3 b) {
    switch (KP) {
        SGL_NNLS_CASE(2, false); SGL_NNLS_CASE(4, false); SGL_NNLS_CASE(6, false); SGL_NNLS_CASE(8, false); SGL_NNLS_CASE(10, false); SGL_NNLS_CASE(12, false);
        SGL_NNLS_CASE(14, false); SGL_NNLS_CASE(16, false); SGL_NNLS_CASE(18, false); SGL_NNLS_CASE(20, false); SGL_NNLS_CASE(22, false); SGL_NNLS_CASE(24, false);
        SGL_NNLS_CASE(26, false); SGL_NNLS_CASE(28, false); SGL_NNLS_CASE(30, false); SGL_NNLS_CASE(32, false); SGL_NNLS_CASE(34, false); SGL_NNLS_CASE(36, false);
        SGL_NNLS_CASE(38, false); SGL_NNLS_CASE(40, false);
        default: sgl_set_error("k_nnls_lane: unsupported KP=%d", KP); return SGL_EINVAL;
    }
    return SGL_OK;
}
