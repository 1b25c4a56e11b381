// nnls (src/singlet.cpp:229-250), one lane per column, with the SWEEP as generated, hand-scheduled assembly
// (gen_nnls_lane.py -> nnls_lane_gen.inc; round 5).  Same interface, same passes / packing protocol (NnlsPass) and the same
// arithmetic in the same order per column as nnls_lane_kernel<KP, true> (nnls_lane.h): bit-identical results.  What differs
// is the schedule inside a sweep: the serial chain of a coordinate step is interleaved with the row-update FMAs of its
// neighbours (header of gen_nnls_lane.py), which the compiler-scheduled kernel leaves to the chance overlap of the two waves
// of a SIMD.  b, x and the sweep's working set live in v[V_T : 255]: all of a column's solve -- loads, sweep loop, stores -- is
// ONE asm statement whose clobbers they are (hipcc cannot be kept out of a register range below 64 ACROSS statements:
// amdgpu_num_vgpr is not honoured, waves_per_eu caps at 64), the compiler keeps v[0 : V_T - 1] for the statement's operands.
#include "sgl_internal.h"
#include "nnls_lane_gen.inc"

#define SGL_DEFINE_NNLS_ASM_KERNEL(KP)                                                                                              \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NNLS_ASM_WAVES_##KP, NNLS_ASM_WAVES_##KP))) void nnls_lane_asm_kernel_##KP(            \
        const double* __restrict__ Gpad, int gs_in, double* __restrict__ B, double* __restrict__ X,                                \
        const int64_t* __restrict__ col_nnz, int k, int64_t ncols, double L1, double L2,                                           \
        unsigned long long* __restrict__ sweep_counter, NnlsPass ps) {                                                             \
        constexpr int NG = (KP + 15) / 16, NGP = NNLS_ASM_NGP_##KP, ROW = 16 * NGP;                                                \
        const int64_t n_in = ps.list ? (int64_t)*ps.count : ncols;                                                                  \
        /* A packed single pass (columns in descending order of their previous sweep counts) of 257 ... 512 workgroups is ONE      */ \
        /* round of the chip, two workgroups per CU, and lasts as long as the SIMD whose two waves take longest: in launch order   */ \
        /* CU c would get the sorted workgroups c and c + 256 -- the longest with the next longest.  A SIMD's time is about         */ \
        /* 7.5 max + 2.1 min sweep units (two waves share the FP64 pipe at 4.8 cycles per instruction, one alone gets 7.5), so the  */ \
        /* second layer runs in ASCENDING order, and the CUs that get no second workgroup (512 - count of them, the END of the       */ \
        /* first layer) host the LONGEST ones: a wave alone takes 7.5 max.  Positions only: a column's arithmetic is the same.       */ \
        unsigned bx = blockIdx.x;                                                                                                   \
        if (ps.fresh == 2 && gridDim.x > 256u && gridDim.x <= 512u) {                                                               \
            const unsigned alone = 512u - gridDim.x;                                                                                \
            bx = bx >= 256u ? gridDim.x - 1u - (bx - 256u) : (bx < 256u - alone ? alone + bx : bx - (256u - alone));               \
        }                                                                                                                           \
        if ((int64_t)bx * blockDim.x >= n_in) return;                                                                               \
        extern __shared__ __attribute__((aligned(16))) double nnls_asm_lds[];                                                       \
        double* const Gl = nnls_asm_lds;              /* Gl[i][l][m] = G[i, l + 16 m] */                                            \
        double* const Dl = nnls_asm_lds + KP * ROW;   /* (G_ii, 1 / G_ii) */                                                        \
        for (int e = threadIdx.x; e < KP * ROW; e += blockDim.x) {                                                                  \
            const int i = e / ROW, r = e - i * ROW, l = r / NGP, m = r - l * NGP, j = l + 16 * m;                                   \
            Gl[e] = (m < NG && j < KP) ? Gpad[j + gs_in * i] : 0.0;                                                                 \
        }                                                                                                                           \
        for (int j = threadIdx.x; j < KP; j += blockDim.x) {                                                                        \
            Dl[2 * j] = Gpad[j * gs_in + j];                                                                                        \
            Dl[2 * j + 1] = Gpad[KP * gs_in + j];   /* row KP of the padded Gram: the correctly rounded reciprocals */              \
        }                                                                                                                           \
        __syncthreads();                                                                                                            \
        const int64_t gid = (int64_t)bx * blockDim.x + threadIdx.x;                                                                 \
        const bool in_range = gid < n_in;                                                                                           \
        const int64_t col = in_range ? (ps.list ? (int64_t)ps.list[gid] : gid) : 0;                                                 \
        const bool resume = ps.list != nullptr && !ps.fresh;                                                                        \
        const bool valid = in_range && (resume || col_nnz == nullptr || col_nnz[col] != 0);                                         \
        const bool to_end = (ps.next_list == nullptr) || n_in <= (int64_t)ps.final_below;                                           \
        double* const bp = B + col * k;                                                                                             \
        double* const xp = X + col * k;                                                                                             \
        typedef __attribute__((address_space(3))) char lds_char;                                                                    \
        const unsigned gl = (unsigned)(uintptr_t)(lds_char*)Gl + (unsigned)(threadIdx.x & 15) * (NGP * 8);                          \
        const unsigned dl = (unsigned)(uintptr_t)(lds_char*)Dl;                                                                     \
        const double kd = (double)k;                                                                                                \
        double tol = 1.0;                                                                                                           \
        int it = 0;                                                                                                                 \
        if (valid && resume) {                                                                                                      \
            tol = ps.tol_state[col];                                                                                                \
            it = (int)ps.it_state[col];                                                                                             \
        }                                                                                                                           \
        /* the column's whole solve -- load b and x, sweep until every lane has stopped (or the pass re-packs), store x and the      \
           b of unfinished columns -- is ONE statement: its registers v[V_T : 255] are clobbers, nothing lives in them outside */   \
        int ran = 0, tlo = __double2loint(tol), thi = __double2hiint(tol);                                                          \
        unsigned long long um = 0ull;                                                                                               \
        const unsigned long long vm = __ballot(valid);                                                                              \
        const double eps = 1e-15, thr = 1e-8;                                                                                       \
        const unsigned one_hi = 0x3ff00000u;                                                                                        \
        const int toend_s = __builtin_amdgcn_readfirstlane(to_end ? 1 : 0), klast_s = __builtin_amdgcn_readfirstlane(k == KP ? 1 : 0); \
        asm volatile(NNLS_ASM_BODY_##KP                                                                                             \
                     : [it] "+v"(it), [lo] "+v"(tlo), [hi] "+v"(thi), [ran] "+s"(ran), [um] "=s"(um)                                \
                     : [bp] "v"(bp), [xp] "v"(xp), [gl] "v"(gl), [dl] "v"(dl), [one_hi] "v"(one_hi), [valid] "s"(vm),               \
                       [l1] "s"(L1), [l2] "s"(L2), [eps] "s"(eps), [kd] "s"(kd), [thr] "s"(thr), [toend] "s"(toend_s),             \
                       [klast] "s"(klast_s)                                                                                         \
                     : NNLS_ASM_VCLOB_##KP, "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",     \
                       "vcc", "scc", "memory");                                                                                     \
        tol = __hiloint2double(thi, tlo);                                                                                           \
        const bool unfinished = ((um >> (threadIdx.x & 63)) & 1ull) != 0ull; /* only possible when !to_end */                       \
        if (unfinished) {                                                                                                           \
            ps.tol_state[col] = tol;                                                                                                \
            ps.it_state[col] = (uint8_t)it;                                                                                         \
        }                                                                                                                           \
        if (valid && !unfinished && ps.prev_it != nullptr) ps.prev_it[col] = (uint8_t)it; /* packing key of the next solve */       \
        const unsigned long long um2 = __ballot(unfinished);                                                                        \
        if (um2 != 0ull) { /* wave-aggregated append */                                                                             \
            const int lane = threadIdx.x & 63;                                                                                      \
            unsigned base = 0;                                                                                                      \
            if (lane == 0) base = atomicAdd(ps.next_count, (unsigned)__popcll(um2));                                                \
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);                                                             \
            if (unfinished) ps.next_list[base + (unsigned)__popcll(um2 & ((1ull << lane) - 1ull))] = (int32_t)col;                  \
        }                                                                                                                           \
        if (sweep_counter != nullptr) {                                                                                             \
            int s = (valid && !unfinished) ? it : 0; /* a column's sweeps are booked once, when it stops */                         \
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);                                                    \
            if ((threadIdx.x & 63) == 0 && (s != 0 || ran != 0)) {                                                                  \
                atomicAdd(sweep_counter, (unsigned long long)s);                                                                    \
                atomicAdd(sweep_counter + 2, (unsigned long long)ran); /* sweeps this wave actually executed */                     \
            }                                                                                                                       \
        }                                                                                                                           \
    }

SGL_NNLS_ASM_INSTANCES(SGL_DEFINE_NNLS_ASM_KERNEL)

// whether the generated sweep serves this padded rank (a padded coordinate is an exact no-op only with L1 >= 0).  Above KP = 50
// (x in the accumulator registers, ONE wave per SIMD) it wins where the solve is short of waves anyway -- the W side's 30 000
// genes: nnls_w 0.596 -> 0.459 ms at k = 64 -- and loses to the compiled kernel's two waves per SIMD on long launches (nnls_h per
// 200 000 cells k = 52: 1.51 -> 1.74 ms, k = 64: 2.56 -> 2.61): used up to 65 536 columns there (profiles/r5_nnls_generated_sweep.txt).
bool nnls_lane_asm_has(int KP, double L1, int64_t ncols) {
    if (!(L1 >= 0.0) || getenv("SGL_NNLS_NO_ASM")) return false;
    if (KP > 50 && ncols > 65536 && !getenv("SGL_NNLS_ASM_ALWAYS")) return false;
#define SGL_NNLS_ASM_HAS(K_) if (KP == K_) return true;
    SGL_NNLS_ASM_INSTANCES(SGL_NNLS_ASM_HAS)
#undef SGL_NNLS_ASM_HAS
    return false;
}

int k_nnls_lane_launch_asm(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz, int k,
                           int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsPass& ps, dim3 g, dim3 b) {
    const int gs_in = nnls_gram_stride(KP);
#define SGL_NNLS_ASM_CASE(K_)                                                                                                       \
    if (KP == K_) {                                                                                                                 \
        const size_t lds = sizeof(double) * ((size_t)K_ * 16 * NNLS_ASM_NGP_##K_ + 2 * K_);                                         \
        nnls_lane_asm_kernel_##K_<<<g, b, lds, s>>>(Gpad, gs_in, B, X, col_nnz, k, ncols, L1, L2, sweep_counter, ps);               \
        HIPCHK(hipGetLastError());                                                                                                  \
        return SGL_OK;                                                                                                              \
    }
    SGL_NNLS_ASM_INSTANCES(SGL_NNLS_ASM_CASE)
#undef SGL_NNLS_ASM_CASE
    sgl_set_error("k_nnls_lane_asm: no instance for KP=%d", KP);
    return SGL_EINVAL;
}
