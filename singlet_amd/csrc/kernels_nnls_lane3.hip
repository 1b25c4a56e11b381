// nnls_lane_kernel<KP, true, true> instances for 64 < k <= 104 (KP in steps of 8; x in a memory scratch,
// see nnls_lane.h)
#include "nnls_lane.h"

int k_nnls_lane_launch3(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                        int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g,
                        dim3 b) {
    switch (KP) {
        SGL_NNLS_CASE_XM(72); SGL_NNLS_CASE_XM(80); SGL_NNLS_CASE_XM(88); SGL_NNLS_CASE_XM(96); SGL_NNLS_CASE_XM(104);
        default: sgl_set_error("k_nnls_lane: unsupported KP=%d", KP); return SGL_EINVAL;
    }
    return SGL_OK;
}
