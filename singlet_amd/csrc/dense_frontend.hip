// Dense front-end (c_nmf_dense / c_ard_nmf_dense, src/singlet.cpp:1052-1054, 1357-1361; dense predict :370-381):
//  * the dense matrix goes to the device as it is and its CSC image (zeros dropped) is built THERE (count / scan /
//    fill with wave-level compaction) -- round 2 built it on the host, one push_back per entry;
//  * when more than half of the entries are non-zero the right-hand sides of predict are what the reference writes,
//    `w * A.col(i)` for every column -- one FP64 GEMM per half-iteration on the matrix cores (rocBLAS, bound at run
//    time like RCCL: a plain library GEMM, MI355X-first rule "hipBLASLt / rocBLAS only for plain library GEMMs") --
//    instead of a sparse accumulate over an image that is not sparse.  Every other step (Gram, NNLS, scale, cor, the
//    masked path's hashing) runs on the same kernels as the sparse fit.
#include "sgl_internal.h"

#include <dlfcn.h>
#include <mutex>
#include <stdlib.h>

// ---- CSC image of a dense column-major matrix ------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_count_kernel(const double* __restrict__ A, int32_t nrow, int64_t ncol,
                                                          int64_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t c = wave; c < ncol; c += nwaves) {
        const double* col = A + (size_t)c * nrow;
        int64_t n = 0;
        for (int64_t r0 = 0; r0 < nrow; r0 += 64) {
            const int64_t r = r0 + lane;
            n += __popcll(__ballot(r < nrow && col[r] != 0.0));
        }
        if (lane == 0) counts[c] = n;
    }
}

__global__ __launch_bounds__(256) void dense_fill_kernel(const double* __restrict__ A, int32_t nrow, int64_t ncol,
                                                         const int64_t* __restrict__ p, int32_t* __restrict__ idx,
                                                         double* __restrict__ x) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t c = wave; c < ncol; c += nwaves) {
        const double* col = A + (size_t)c * nrow;
        int64_t pos = p[c];
        for (int64_t r0 = 0; r0 < nrow; r0 += 64) {
            const int64_t r = r0 + lane;
            const double v = (r < nrow) ? col[r] : 0.0;
            const unsigned long long m = __ballot(v != 0.0);
            if (v != 0.0) {
                const int64_t dst = pos + __popcll(m & ((1ull << lane) - 1ull));
                idx[dst] = (int32_t)r;
                x[dst] = v;
            }
            pos += __popcll(m);
        }
    }
}

int k_dense_count(hipStream_t s, const double* A, int32_t nrow, int64_t ncol, int64_t* counts) {
    if (ncol <= 0) return SGL_OK;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((ncol + 3) / 4, 256 * 16));
    dense_count_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(A, nrow, ncol, counts);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

int k_dense_fill(hipStream_t s, const double* A, int32_t nrow, int64_t ncol, const int64_t* p, int32_t* idx, double* x) {
    if (ncol <= 0) return SGL_OK;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((ncol + 3) / 4, 256 * 16));
    dense_fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(A, nrow, ncol, p, idx, x);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// out[e] = part[0][e] + part[1][e] + ... in slab order
__global__ void dense_sum_slabs_kernel(const double* __restrict__ part, int R, int64_t n, double* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        double s = part[e];
        for (int r = 1; r < R; ++r) s += part[(size_t)r * n + e];
        out[e] = s;
    }
}
static int k_sum_slabs(hipStream_t s, const double* part, int R, int64_t n, double* out) {
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096));
    dense_sum_slabs_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(part, R, n, out);
    HIPCHK(hipGetLastError());
    return SGL_OK;
}

// ---- rocBLAS, bound at run time ----------------------------------------------------------------------------
typedef struct _rocblas_handle* rb_handle;
struct RocblasApi {
    void* lib = nullptr;
    int (*create_handle)(rb_handle*) = nullptr;
    int (*destroy_handle)(rb_handle) = nullptr;
    int (*set_stream)(rb_handle, hipStream_t) = nullptr;
    int (*set_atomics_mode)(rb_handle, int) = nullptr;   // optional: rocblas_atomics_not_allowed = 0
    int (*dgemm)(rb_handle, int, int, int, int, int, const double*, const double*, int, const double*, int, const double*, double*,
                 int) = nullptr;
    int (*dgemm_sb)(rb_handle, int, int, int, int, int, const double*, const double*, int, long long, const double*, int, long long,
                    const double*, double*, int, long long, int) = nullptr;
};

static RocblasApi* rocblas_api() {
    static RocblasApi api;
    static char why[256] = "not found";
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("SGL_ROCBLAS_PATH"), "librocblas.so.5", "librocblas.so.4", "librocblas.so", "/opt/rocm/lib/librocblas.so.5",
                               "/opt/rocm/lib/librocblas.so.4", "/opt/rocm/lib/librocblas.so"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
            const char* e = dlerror();
            if (e) snprintf(why, sizeof(why), "%s", e);
        }
        if (api.lib) {
            bool ok = true;
            auto bind = [&](const char* sym) { void* f = dlsym(api.lib, sym); if (!f) { ok = false; snprintf(why, sizeof(why), "symbol %s missing", sym); } return f; };
            api.create_handle = (decltype(api.create_handle))bind("rocblas_create_handle");
            api.destroy_handle = (decltype(api.destroy_handle))bind("rocblas_destroy_handle");
            api.set_stream = (decltype(api.set_stream))bind("rocblas_set_stream");
            api.dgemm = (decltype(api.dgemm))bind("rocblas_dgemm");
            api.dgemm_sb = (decltype(api.dgemm_sb))bind("rocblas_dgemm_strided_batched");
            if (!ok) { dlclose(api.lib); api.lib = nullptr; }
            else api.set_atomics_mode = (decltype(api.set_atomics_mode))dlsym(api.lib, "rocblas_set_atomics_mode");
        }
    });
    if (!api.lib) { sgl_set_error("rocBLAS (librocblas.so) could not be loaded: %s", why); return nullptr; }
    return &api;
}

// can the GEMM path run at all?  Asked once at sgl_upload_dense: without rocBLAS the dense fit keeps the CSC image
// (the path of round 2, slower on dense data but always there) instead of failing in the first iteration
bool sgl_dense_gemm_available() { return rocblas_api() != nullptr; }

void sgl_dense_release(sgl_ctx* c) {
    if (c->rocblas) {
        RocblasApi* R = rocblas_api();
        if (R) (void)R->destroy_handle((rb_handle)c->rocblas);
        c->rocblas = nullptr;
    }
    if (c->Adense) (void)sgl_pool_free(c->Adense);
    c->Adense = nullptr;
    c->dense_gemm = false;
}

// which = 0: B (k x ncol) = F (k x nrow) * A;   which = 1: B (k x nrow) = F (k x ncol) * A^T.   A: nrow x ncol, column-major.
int k_dense_rhs(sgl_ctx* c, int which, const double* F, int k, double* B) {
    RocblasApi* R = rocblas_api();
    if (!R) return SGL_ECOMM;
    if (!c->rocblas) {
        rb_handle h = nullptr;
        if (R->create_handle(&h) != 0) { sgl_set_error("rocblas_create_handle failed"); return SGL_EHIP; }
        // no atomics inside the GEMMs (split-K sums with atomics are not reproducible run to run; everything else in the
        // library sums in a fixed order).  The default of recent rocBLAS releases already is "not allowed".
        if (R->set_atomics_mode) (void)R->set_atomics_mode(h, 0);
        c->rocblas = h;
    }
    rb_handle h = (rb_handle)c->rocblas;
    if (R->set_stream(h, c->stream) != 0) { sgl_set_error("rocblas_set_stream failed"); return SGL_EHIP; }
    const double one = 1.0, zero = 0.0;
    const int m = c->A.nrow, n = c->A.ncol;
    if (which == 0) {
        const int rc = R->dgemm(h, 111, 111, k, n, m, &one, F, k, c->Adense, m, &zero, B, k);
        if (rc != 0) { sgl_set_error("rocblas_dgemm failed with status %d", rc); return SGL_EHIP; }
        return SGL_OK;
    }
    // W side: a k x m result over a contraction as long as the cells.  One GEMM leaves the chip to k * m / tile
    // workgroups (2.7 ms at 2000 x 50 000, k = 50; 5.4 ms on a kept t(A)); the cells are cut into P slices, one
    // strided-batched GEMM forms the P partial products, and they are summed in slice order (fixed: reproducible).
    int P = (int)std::min<int64_t>(64, std::max<int64_t>(1, n / 512));
    const int Kc = n / P, tail = n - Kc * P;
    const size_t per = (size_t)k * m;
    SGLCHK(sgl_ws_reserve(c, sizeof(double) * per * (size_t)(P + 1)));
    int rc = R->dgemm_sb(h, 111, 112, k, m, Kc, &one, F, k, (long long)k * Kc, c->Adense, m, (long long)m * Kc, &zero, c->ws, k,
                         (long long)per, P);
    if (rc == 0 && tail > 0)
        rc = R->dgemm(h, 111, 112, k, m, tail, &one, F + (size_t)k * Kc * P, k, c->Adense + (size_t)m * Kc * P, m, &zero, c->ws + per * P, k);
    if (rc != 0) { sgl_set_error("rocblas_dgemm_strided_batched failed with status %d", rc); return SGL_EHIP; }
    return k_sum_slabs(c->stream, c->ws, P + (tail > 0 ? 1 : 0), (int64_t)per, B);
}
