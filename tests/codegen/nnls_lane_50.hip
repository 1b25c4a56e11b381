// One instance of the lane-per-column NNLS kernel (vector-load + DPP operand path), compiled to assembly by
// tests/test_kernel_codegen.py to check the hand-written DPP instructions against hazards hipcc cannot see.
#include "nnls_lane.h"
template __global__ void nnls_lane_kernel<50, true, false>(const double*, double*, double*, const int64_t*, int, int64_t, double,
                                                    double, unsigned long long*, NnlsPass);
void sgl_set_error(const char*, ...) {}
