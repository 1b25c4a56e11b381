// nnls_lane_kernel<KP> instances for KP = 42 .. 64 (see nnls_lane.h)
#include "nnls_lane.h"

int k_nnls_lane_launch2(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                        int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g,
                        dim3 b) {
    switch (KP) {
        SGL_NNLS_CASE(42); SGL_NNLS_CASE(44); SGL_NNLS_CASE(46); SGL_NNLS_CASE(48); SGL_NNLS_CASE(50); SGL_NNLS_CASE(52);
        SGL_NNLS_CASE(54); SGL_NNLS_CASE(56); SGL_NNLS_CASE(58); SGL_NNLS_CASE(60); SGL_NNLS_CASE(62); SGL_NNLS_CASE(64);
        default: sgl_set_error("k_nnls_lane: unsupported KP=%d", KP); return SGL_EINVAL;
    }
    return SGL_OK;
}
