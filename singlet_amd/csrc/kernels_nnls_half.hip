// nnls_half_kernel<KH> instances for 64 < k <= 128 (two lanes per column, see nnls_half.h)
#include "nnls_half.h"
#include <atomic>

template <int KH>
static int launch_half(dim3 g, dim3 b, hipStream_t s, const double* Gpad, double* B, double* X, const int64_t* col_nnz, int k,
                       int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps) {
    constexpr size_t lds = nnls_half_lds_bytes<KH>();
    static std::atomic<bool> attr_set[64];   // per (instance, device)
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nnls_half_kernel<KH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    nnls_half_kernel<KH><<<g, b, lds, s>>>(Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);
    HIPCHK(hipGetLastError());   // up to 131 KB of dynamic LDS: a refused launch must not leave X stale unnoticed
    return SGL_OK;
}

// KP = 72 ... 104 in steps of 8.  Workgroups of 256 threads (128 columns); where the staged Gram leaves room for only one
// workgroup per CU (KP = 104: 94 KB of LDS) and there are columns enough, 512 threads, so that a SIMD still has two waves.
int k_nnls_half_launch(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                       int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps) {
    const bool big = KP > 96 && KP <= 104 && ncols >= 256 * 256;   // (above 104: x in AGPRs, one wave per SIMD, 256 threads)
    const int cpb = big ? 256 : 128;
    const dim3 g((unsigned)((ncols + cpb - 1) / cpb)), b(2 * cpb);
    switch (KP) {
        case 72: return launch_half<36>(g, b, s, Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);
        case 80: return launch_half<40>(g, b, s, Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);
        case 88: return launch_half<44>(g, b, s, Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);
        case 96: return launch_half<48>(g, b, s, Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);
        case 104: return launch_half<52>(g, b, s, Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);
        case 112: return launch_half<56>(g, b, s, Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);
        case 120: return launch_half<60>(g, b, s, Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);
        case 128: return launch_half<64>(g, b, s, Gpad, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);
        default: sgl_set_error("k_nnls_half: unsupported KP=%d", KP); return SGL_EINVAL;
    }
}
