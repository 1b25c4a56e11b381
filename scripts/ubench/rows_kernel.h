// Row-shared sparse accumulate:  B[:, c] = sum_{(r, v) in column c} v * F[:, r]   (predict, src/singlet.cpp:341-343)
//
// The column-major LDS-tiled kernel (kernels_tiled.hip) reads the k-vector F[:, r] from LDS once per non-zero;
// this one walks a (64-column block, row tile) chunk ROW-major and reads F[:, r] once per *part* = one row of the
// chunk with up to two of its non-zeros, lane = factor (ds_read_b64: 2 LDS cycles per part).  The 64 column
// accumulators of a wave stay in v[128:255] for the whole pass; the entry's column selects its accumulator through
// M0-relative VGPR addressing of the FMA's destination (one SALU instruction per entry, written straight from the
// packed column word).  Inner loop and stream format: gen_rows.py.  NOT part of libsinglet_hip.so: built, verified and
// measured on MI355X at config-3 shape (rows_bench.hip; profiles/r2_rows_kernel.md) -- 11.8 ms per pass against 12.6 ms
// of the column-major kernel, not enough to carry a second stream format in the product.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rows_gen.inc"

#define ROWS_NW 8                             // waves per workgroup (one workgroup per CU: the tile takes the LDS)
#define ROWS_CW 64                            // columns per wave
#define ROWS_LDS_BYTES (160 * 1024 - 512)     // tile bytes; the 512 B of slack are read (never used) by lanes >= k of the last row

#ifndef ROWS_CHUNK_ASM
#define ROWS_CHUNK_ASM ACC_ROWS_CHUNK_ASM
#endif

__device__ __forceinline__ void rows_acc_load(int idx2, double& v) {
    int a, b;
    asm volatile("s_set_gpr_idx_on %2, gpr_idx(SRC0)\n\tv_mov_b32 %0, v128\n\tv_mov_b32 %1, v129\n\ts_set_gpr_idx_off"
                 : "=v"(a), "=v"(b) : "s"(idx2) : "memory");
    v = __hiloint2double(b, a);
}

// stream: records of ACC_ROWS_REC_BYTES; bstart[wb * T + t] = first record of chunk (wb, t); nblk = records of the chunk
__global__ __launch_bounds__(64 * ROWS_NW) __attribute__((amdgpu_num_vgpr(64), amdgpu_waves_per_eu(8, 8))) void acc_rows_kernel(
    const char* __restrict__ stream, const int64_t* __restrict__ bstart, const uint16_t* __restrict__ nblk, int T, int64_t nwb,
    const double* __restrict__ F, int k, int TR, int64_t nrow, int tiles_per_range, double* __restrict__ Bout, int64_t ncol, int ldf,
    int ldb, int64_t slab) {
    // The compiler's budget is v0..v63 (waves_per_eu(8, 8) caps its allocation at 512 / 8 registers); the clobber makes
    // the kernel descriptor allocate all 256: v64..v255 belong to the inline asm (ring loads stay in flight across
    // compiler code, so these registers must never be touched by it).
    asm volatile("" ::: "v255");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wb = (int64_t)blockIdx.x * ROWS_NW + wave;
    const int t0 = blockIdx.y * tiles_per_range;
    const int t1 = (t0 + tiles_per_range < T) ? (t0 + tiles_per_range) : T;
    const bool wact = wb < nwb;
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lane8 = (unsigned)(uintptr_t)(lds_char*)smem + lane * 8;
    const unsigned xoff = lane * 8, coff = lane * 4;

    asm volatile(ACC_ROWS_ZERO_ASM ::: "memory");

    uint64_t ptr = 0;
    int phase = 0;
    if (wact) {
        const int64_t b0 = bstart[wb * T + t0];
        const uint64_t p = reinterpret_cast<uint64_t>(stream) + (uint64_t)b0 * ACC_ROWS_REC_BYTES;
        ptr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((unsigned)p);
        asm volatile(ACC_ROWS_RING_FILL_ASM : [ptr] "+s"(ptr) : [xoff] "v"(xoff), [coff] "v"(coff) : ACC_ROWS_CLOBBERS);
    }

    for (int t = t0; t < t1; ++t) {
        int nb = 0;
        if (wact) nb = __builtin_amdgcn_readfirstlane((int)nblk[wb * T + t]);
        // stage rows [t * TR, ...) of F into LDS, row stride k doubles
        const int64_t row0 = (int64_t)t * TR;
        const int rows = (int)((nrow - row0 < TR) ? (nrow - row0) : TR);
        const int n = rows * k;
        const double* __restrict__ src = F + row0 * ldf;
#ifdef ROWS_NOSTAGE
        if (t > t0) { __syncthreads(); } else
#endif
        if (ldf == k && (n & 1) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            // contiguous tile: every thread moves NRND * RST 16-byte pieces, RST loads in flight; the first round
            // is issued before the barrier and overlaps the other waves' tail of the previous tile
            constexpr int RST = 8;
            constexpr int NRND = (ROWS_LDS_BYTES / 16 + RST * 64 * ROWS_NW - 1) / (RST * 64 * ROWS_NW);
            double2 stg[RST];
#pragma unroll
            for (int j = 0; j < RST; ++j) {
                const int e = ((int)threadIdx.x + j * 64 * ROWS_NW) * 2;
                stg[j] = double2{0.0, 0.0};
                if (e < n) stg[j] = *reinterpret_cast<const double2*>(src + e);
            }
            __syncthreads();  // everyone is done reading the previous tile
#pragma unroll
            for (int rd = 0; rd < NRND; ++rd) {
#pragma unroll
                for (int j = 0; j < RST; ++j) {
                    const int e = ((int)threadIdx.x + (rd * RST + j) * 64 * ROWS_NW) * 2;
                    if (e < n) *reinterpret_cast<double2*>(tile + e) = stg[j];
                }
                if (rd + 1 < NRND) {
#pragma unroll
                    for (int j = 0; j < RST; ++j) {
                        const int e = ((int)threadIdx.x + ((rd + 1) * RST + j) * 64 * ROWS_NW) * 2;
                        stg[j] = double2{0.0, 0.0};
                        if (e < n) stg[j] = *reinterpret_cast<const double2*>(src + e);
                    }
                }
            }
        } else {
            __syncthreads();
            for (int e = (int)threadIdx.x; e < n; e += 64 * ROWS_NW) {
                const int r = e / k, f = e - r * k;
                tile[e] = src[(int64_t)r * ldf + f];
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see kernels_tiled.hip
        __syncthreads();
        if (nb > 0) {
            asm volatile(ROWS_CHUNK_ASM
                         : [ptr] "+s"(ptr), [phase] "+s"(phase)
                         : [nb] "s"(nb), [lane8] "v"(lane8), [xoff] "v"(xoff), [coff] "v"(coff)
                         : ACC_ROWS_CLOBBERS);
        }
    }
    // refills issued past the end of this wave's range are still in flight: they write the ring registers only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wact) {
        double* out = Bout + (size_t)blockIdx.y * (size_t)slab;
        for (int c = 0; c < ROWS_CW; ++c) {
            double v;
            rows_acc_load(2 * c, v);
            const int64_t col = wb * ROWS_CW + c;
            if (col < ncol && lane < k) out[col * ldb + lane] = v;
        }
    }
}
