// nnls_lane_kernel<KP> instances for KP = 42 .. 64 (see nnls_lane.h)
#include "nnls_lane.h"

int k_nnls_lane_launch2(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                        int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g,
                        dim3 b) {
    switch (KP) {
        SGL_NNLS_CASE(42, true); SGL_NNLS_CASE(44, true); SGL_NNLS_CASE(46, true); SGL_NNLS_CASE(48, true); SGL_NNLS_CASE(50, true); SGL_NNLS_CASE(52, true);
        SGL_NNLS_CASE(54, true); SGL_NNLS_CASE(56, true); SGL_NNLS_CASE(58, true); SGL_NNLS_CASE(60, true); SGL_NNLS_CASE(62, true); SGL_NNLS_CASE(64, true);
        default: sgl_set_error("k_nnls_lane: unsupported KP=%d", KP); return SGL_EINVAL;
    }
    return SGL_OK;
}
