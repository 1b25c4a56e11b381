// nnls_lane_kernel<KP, true, true> instances for 104 < k <= 128 (KP in steps of 8; x in a memory scratch,
// see nnls_lane.h).  b alone takes 2 KP >= 224 VGPRs here, so these instances are built for ONE wave per
// SIMD (512 registers): slower per SIMD than the k <= 104 instances, but an order of magnitude ahead of
// the wave-per-column fallback they replace (k = 120 on 200 k cells: 140 ms there).
#define SGL_NNLS_WPE 1
#include "nnls_lane.h"

int k_nnls_lane_launch4(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                        int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g,
                        dim3 b) {
    switch (KP) {
        SGL_NNLS_CASE_XM(112); SGL_NNLS_CASE_XM(120); SGL_NNLS_CASE_XM(128);
        default: sgl_set_error("k_nnls_lane: unsupported KP=%d", KP); return SGL_EINVAL;
    }
    return SGL_OK;
}
