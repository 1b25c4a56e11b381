// nnls (src/singlet.cpp:229-250) for 64 < k <= 128, TWO lanes per column, with the whole solve as generated, hand-scheduled
// assembly (gen_nnls_half.py -> nnls_half_gen.inc; round 5).  Same interface, passes / packing protocol (NnlsPass) and results as
// nnls_half_kernel<KH> (nnls_half.h): lane c and lane 32 + c of a wave share column c of the wave's 32, the lower half-wave holds
// coordinates 0 .. KH - 1 of b and x, the upper half KH .. 2 KH - 1.  What differs is inside the sweep (header of gen_nnls_half.py):
// one half computes a coordinate's step, only nd crosses the halves, and the step's chain is interleaved with the row-update FMAs
// of its neighbours.  As in kernels_nnls_asm.hip the solve is ONE asm statement whose clobbers b, x and its working set are.
#include "sgl_internal.h"
#include "nnls_half_gen.inc"
#include <atomic>

#define SGL_DEFINE_NNLS_HALF_ASM_KERNEL(KP)                                                                                         \
    __global__ __launch_bounds__(NNLS_HALF_ASM_THREADS_##KP) __attribute__((amdgpu_waves_per_eu(NNLS_HALF_ASM_WAVES_##KP, NNLS_HALF_ASM_WAVES_##KP)))      \
    void nnls_half_asm_kernel_##KP(const double* __restrict__ Gpad, int gs_in, double* __restrict__ B, double* __restrict__ X,       \
                                   const int64_t* __restrict__ col_nnz, int k, int64_t ncols, double L1, double L2,                 \
                                   unsigned long long* __restrict__ sweep_counter, NnlsPass ps) {                                   \
        constexpr int KH = KP / 2, NGH = NNLS_HALF_ASM_NGH_##KP, NGHP = NNLS_HALF_ASM_NGHP_##KP, ROW = 32 * NGHP;                   \
        const int64_t n_in = ps.list ? (int64_t)*ps.count : ncols;                                                                  \
        const int cpb = (int)(blockDim.x >> 1); /* columns per workgroup */                                                         \
        if ((int64_t)blockIdx.x * cpb >= n_in) return;                                                                              \
        extern __shared__ __attribute__((aligned(16))) double nnls_half_asm_lds[];                                                  \
        double* const Dl = nnls_half_asm_lds;            /* (G_ii, 1 / G_ii) */                                                     \
        double* const Gl = nnls_half_asm_lds + 2 * KP;   /* Gl[i][h][l][m] = G[i, h KH + l + 16 m] */                               \
        for (int e = threadIdx.x; e < KP * ROW; e += blockDim.x) {                                                                  \
            const int i = e / ROW, r = e - i * ROW, hl = r / NGHP, m = r - hl * NGHP, h = hl >> 4, jl = (hl & 15) + 16 * m;          \
            Gl[e] = (m < NGH && jl < KH) ? Gpad[h * KH + jl + gs_in * i] : 0.0;                                                     \
        }                                                                                                                           \
        for (int j = threadIdx.x; j < KP; j += blockDim.x) {                                                                        \
            Dl[2 * j] = Gpad[j * gs_in + j];                                                                                        \
            Dl[2 * j + 1] = Gpad[KP * gs_in + j];   /* row KP of the padded Gram: the correctly rounded reciprocals */              \
        }                                                                                                                           \
        __syncthreads();                                                                                                            \
        const int lane = threadIdx.x & 63, half = lane >> 5;                                                                        \
        const int64_t gid = (int64_t)blockIdx.x * cpb + (threadIdx.x >> 6) * 32 + (lane & 31);   /* position in this pass */         \
        const bool in_range = gid < n_in;                                                                                           \
        const int64_t col = in_range ? (ps.list ? (int64_t)ps.list[gid] : gid) : 0;                                                 \
        const bool resume = ps.list != nullptr && !ps.fresh;                                                                        \
        const bool valid = in_range && (resume || col_nnz == nullptr || col_nnz[col] != 0);                                         \
        const bool to_end = (ps.next_list == nullptr) || n_in <= (int64_t)ps.final_below;                                           \
        double* const bp = B + col * k + half * KH;                                                                                 \
        double* const xp = X + col * k + half * KH;                                                                                 \
        const unsigned kh = (unsigned)(half ? k - KH : KH);   /* coordinates this lane holds */                                     \
        typedef __attribute__((address_space(3))) char lds_char;                                                                    \
        const unsigned gl = (unsigned)(uintptr_t)(lds_char*)Gl + (unsigned)(half * 16 + (lane & 15)) * (NGHP * 8);                  \
        const unsigned dl = (unsigned)(uintptr_t)(lds_char*)Dl;                                                                     \
        const double kd = (double)k;                                                                                                \
        double tol = 1.0;                                                                                                           \
        int it = 0;                                                                                                                 \
        if (valid && resume) {                                                                                                      \
            tol = ps.tol_state[col];                                                                                                \
            it = (int)ps.it_state[col];                                                                                             \
        }                                                                                                                           \
        int ran = 0, tlo = __double2loint(tol), thi = __double2hiint(tol);                                                          \
        unsigned long long um = 0ull;                                                                                               \
        const unsigned long long vm = __ballot(valid);                                                                              \
        const double eps = 1e-15, thr = 1e-8;                                                                                       \
        const unsigned one_hi = 0x3ff00000u;                                                                                        \
        const int toend_s = __builtin_amdgcn_readfirstlane(to_end ? 1 : 0);                                                         \
        asm volatile(NNLS_HALF_ASM_BODY_##KP                                                                                        \
                     : [it] "+v"(it), [lo] "+v"(tlo), [hi] "+v"(thi), [ran] "+s"(ran), [um] "=s"(um)                                \
                     : [bp] "v"(bp), [xp] "v"(xp), [gl] "v"(gl), [dl] "v"(dl), [one_hi] "v"(one_hi), [kh] "v"(kh), [valid] "s"(vm), \
                       [l1] "s"(L1), [l2] "s"(L2), [eps] "s"(eps), [kd] "s"(kd), [thr] "s"(thr), [toend] "s"(toend_s)               \
                     : NNLS_HALF_ASM_VCLOB_##KP, "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",       \
                       "s51", "vcc", "scc", "memory");                                                                              \
        tol = __hiloint2double(thi, tlo);                                                                                           \
        const bool unfinished = ((um >> lane) & 1ull) != 0ull; /* only possible when !to_end; both lanes of a column agree */       \
        if (unfinished && half == 0) {                                                                                              \
            ps.tol_state[col] = tol;                                                                                                \
            ps.it_state[col] = (uint8_t)it;                                                                                         \
        }                                                                                                                           \
        if (valid && !unfinished && half == 0 && ps.prev_it != nullptr) ps.prev_it[col] = (uint8_t)it; /* packing key of the next solve */ \
        const unsigned long long um2 = __ballot(unfinished) & 0xffffffffull;                                                        \
        if (um2 != 0ull) { /* wave-aggregated append (the lower half-wave speaks for the columns) */                                \
            unsigned base = 0;                                                                                                      \
            if (lane == 0) base = atomicAdd(ps.next_count, (unsigned)__popcll(um2));                                                \
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);                                                             \
            if (unfinished && half == 0) ps.next_list[base + (unsigned)__popcll(um2 & ((1ull << lane) - 1ull))] = (int32_t)col;    \
        }                                                                                                                           \
        if (sweep_counter != nullptr) {                                                                                             \
            int s = (valid && !unfinished && half == 0) ? it : 0; /* a column's sweeps are booked once, when it stops */            \
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);                                                    \
            if (lane == 0 && (s != 0 || ran != 0)) {                                                                                \
                atomicAdd(sweep_counter, (unsigned long long)s);                                                                    \
                atomicAdd(sweep_counter + 2, (unsigned long long)ran); /* sweeps this wave actually executed */                     \
            }                                                                                                                       \
        }                                                                                                                           \
    }

SGL_NNLS_HALF_ASM_INSTANCES(SGL_DEFINE_NNLS_HALF_ASM_KERNEL)

// padded rank of the generated two-lane solve serving rank k (0: none).  Instances are multiples of 4 -- k = 100 runs unpadded,
// where the compiled kernels' multiples of 8 make it 104 -- and need L1 >= 0 (a padded coordinate is an exact no-op only then).
int nnls_half_asm_kp(int k, double L1) {
    if (k <= 64 || !(L1 >= 0.0) || getenv("SGL_NNLS_NO_ASM") || getenv("SGL_NNLS_NO_HALF")) return 0;
    const int KP = (k + 3) / 4 * 4;
#define SGL_NNLS_HALF_ASM_HAS(K_) if (KP == K_) return KP;
    SGL_NNLS_HALF_ASM_INSTANCES(SGL_NNLS_HALF_ASM_HAS)
#undef SGL_NNLS_HALF_ASM_HAS
    return 0;
}

int k_nnls_half_launch_asm(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                           int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps) {
    const int gs_in = nnls_gram_stride(KP);
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
#define SGL_NNLS_HALF_ASM_CASE(K_)                                                                                                  \
    if (KP == K_) {                                                                                                                 \
        const size_t lds = sizeof(double) * ((size_t)K_ * 32 * NNLS_HALF_ASM_NGHP_##K_ + 2 * K_);                                   \
        /* the staged Gram leaves room for ONE workgroup per CU from KP = 80 on: 512 threads there (two waves per SIMD) when     */ \
        /* there are columns enough, 256 below                                                                                   */ \
        const int cpb = (2 * lds > 160 * 1024 && ncols >= 256 * 256 && NNLS_HALF_ASM_WAVES_##K_ > 1) ? 256 : 128;                   \
        const dim3 g((unsigned)((ncols + cpb - 1) / cpb)), b(2 * cpb);                                                              \
        static std::atomic<bool> attr_set[64];                                                                                      \
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {                                                                               \
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nnls_half_asm_kernel_##K_),                                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                     \
            if (dev >= 0 && dev < 64) attr_set[dev] = true;                                                                         \
        }                                                                                                                           \
        nnls_half_asm_kernel_##K_<<<g, b, lds, s>>>(Gpad, gs_in, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);               \
        HIPCHK(hipGetLastError());                                                                                                  \
        return SGL_OK;                                                                                                              \
    }
    SGL_NNLS_HALF_ASM_INSTANCES(SGL_NNLS_HALF_ASM_CASE)
#undef SGL_NNLS_HALF_ASM_CASE
    sgl_set_error("k_nnls_half_asm: no instance for KP=%d", KP);
    return SGL_EINVAL;
}
