// Internal declarations shared by the translation units of libsinglet_hip.so.
// gfx950 (MI355X, CDNA4) only: wave = 64 lanes, no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <vector>
#include <algorithm>

#include "../../include/singlet_hip.h"

#define SGL_WAVE 64
#define SGL_MAX_K 1024         // rank limit of the library: above 128 (plain fit) / 128 (masked fit) the generic kernels run
#define SGL_LANE_NNLS_MAX_K 128  // lane-per-column NNLS: k <= 64 all in registers, k <= 128 with x in a memory scratch

void sgl_set_error(const char* fmt, ...);

// device-memory pool (pool.hip): every allocation of the library goes through these; blocks >= 64 MB are cached on free
hipError_t sgl_pool_malloc_raw(void** p, size_t bytes);
template <typename T_>
inline hipError_t sgl_pool_malloc(T_** p, size_t bytes) { return sgl_pool_malloc_raw(reinterpret_cast<void**>(p), bytes); }
hipError_t sgl_pool_free(void* p);
hipError_t sgl_pool_mem_info(size_t* free_b, size_t* total_b);
void sgl_pool_release(void);
size_t sgl_pool_cached_bytes(void);

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            sgl_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return SGL_EHIP;                                                                           \
        }                                                                                              \
    } while (0)

#define SGLCHK(expr)              \
    do {                          \
        int r__ = (expr);         \
        if (r__ != SGL_OK) return r__; \
    } while (0)

// One orientation of the resident shard: CSC with 64-bit column pointers.
struct DevCSC {
    double* x = nullptr;
    int32_t* i = nullptr;
    int64_t* p = nullptr;
    int32_t nrow = 0, ncol = 0;
    int64_t nnz = 0;
    // Tiling of the row range into L2-sized pieces for the sparse accumulate:
    // seg[t * ncol + c] = first non-zero of column c with row >= t * tile_rows
    // (t = 0..ntiles), so tile t of column c is [seg[t][c], seg[t+1][c]).
    int64_t* seg = nullptr;
    int32_t tile_rows = 0, ntiles = 0;
};

// Re-blocked non-zero stream for the LDS-tiled accumulate (kernels_tiled.hip).
struct DevTiled {
    uint32_t* roff = nullptr;   // byte offset of the row inside the LDS tile, per entry
    double* x = nullptr;        // value per entry (0 for pads)
    int64_t* cstart = nullptr;  // [nwb * T + 1] first entry of chunk (wb, t)
    uint8_t* cnt = nullptr;     // [nwb * T * CW] 4-entry groups of slot s in chunk (wb, t)
    uint16_t* gtab = nullptr;   // [E / (4 NSL) + slack] schedule: the M0 word (0x8000 | 4 * column unit) of every group of 4 entry tuples
    size_t cap_gtab = 0;
    double* part = nullptr;     // [R][k * ncol] partial slabs when the tile range is split
    double* xm = nullptr;       // value per entry with the cross-validation mask applied (0 at drawn entries), per fit
    uint64_t xm_seed = 0, xm_inv = 0;
    int xm_mask_t = -1;
    int64_t* seg = nullptr;     // [(T + 1) * ncol] first non-zero of column c at or below row t * TR (kept for the masked values)
    int32_t* perm = nullptr;    // [ncol] column at position pos of the descending-non-zero-count order (nullptr: matrix order)
    size_t cap_perm = 0;
    int64_t perm_nnz = -1;      // the matrix (by its non-zero count) perm was computed for
    bool range_fastest = false; // work units ordered (column group, tile range) instead of (tile range, column group)
    double top_share = 0.0;     // share of the non-zeros in the first 512 columns of that order (the heaviest workgroup)
    double top_share4 = 0.0;    // ... in the first 1024 (the heaviest workgroup of the quad layout)
    // The buffers outlive a fit: a rank sweep re-inits the fit tens of times on one matrix, and hipMalloc / hipFree of
    // tens of GB cost up to seconds each at config-5 size.  cap_* = allocated element counts; `built` = the stream
    // content is valid for (k, src_nnz); any change of the matrix frees everything (sgl_tiled_free).
    size_t cap_roff = 0, cap_x = 0, cap_cstart = 0, cap_cnt = 0, cap_part = 0, cap_xm = 0, cap_seg = 0;
    bool built = false;
    int64_t builds = 0;         // times the stream content was (re)written on this context (sgl_layout_builds)
    int64_t src_nnz = -1;
    int32_t TR = 0, T = 0, CW = 0, k = 0, R = 1, tiles_per_range = 0;
    int64_t tail_wg0 = -1;      // tail split (R == 1): first workgroup (x) of the last, partly filled round of 256 (-1: none)
    int32_t tail_R = 1;         //   its workgroups' tile ranges are cut into this many pieces (own slabs, summed in order)
    int32_t NSL = 2;            // column slots per LDS instruction: 2 (pairs, k <= 64) or 4 (quads, k <= 32); CW = 32 * NSL
    int32_t KS = 0;             // LDS row stride in doubles the row offsets were built for
    int64_t nwb = 0, E = 0, ncol = 0, nrow = 0;
};

struct PhaseEvent {
    int phase;
    hipEvent_t e0, e1;
};

// one pass of the lane-per-column kernel (nnls_lane.h): columns come from `list` (count at *count)
// or are 0..ncols-1 when list == nullptr; unfinished columns go to next_list (nullptr: run to the end)
struct NnlsPass {
    const int32_t* list;
    const uint32_t* count;
    int32_t* next_list;
    uint32_t* next_count;
    uint8_t* it_state;    // sweeps done so far, per column
    double* tol_state;    // running tol, per column
    int32_t final_below;  // a pass over at most this many columns runs them to the end
    double* xt;           // k > 64: scratch holding x, xt[i * xt_stride + position in this pass]
    int64_t xt_stride;
    int32_t fresh;        // list != nullptr but nothing to resume: a first pass over columns given in packing order (below)
    uint8_t* prev_it;     // sweeps a column needed, written when it stops: the packing key of the NEXT solve (nullptr: not kept)
};
// device scratch of the multi-pass solve, sized for `cap` columns (owned by the caller)
struct NnlsScratch {
    int32_t* list[2] = {nullptr, nullptr};
    uint32_t* counts = nullptr;   // SGL_NNLS_MAX_PASSES + 1
    uint8_t* it_state = nullptr;
    double* tol_state = nullptr;
    double* xt = nullptr;         // k x cap doubles when the fit's rank is above 64
    int64_t cap = 0;
    // packing by sweep count (H side of a plain fit): per column the sweeps of the previous solve, the columns in descending
    // order of it, and the counting sort's workspace
    uint8_t* prev_it = nullptr;
    int32_t* packed = nullptr;
    uint32_t* sort_ws = nullptr;  // [128 * nblocks + 2]
    int64_t pack_cap = 0;
};

// The cross-validation mask of one orientation as lists: idx[ptr[c] .. ptr[c + 1]) = the rows r (ascending) with
// draw(cell, gene) true for column c -- built once per (seed, inv_density, shape, offsets) by two hashing passes and
// read by mask_gram_list_kernel every iteration instead of hashing all rows x columns again (kernels_mask.hip).
#define SGL_ML_KEEP 3
struct DevMaskList {
    int64_t* ptr = nullptr;    // ncol + 1
    int32_t* idx = nullptr;
    double* val = nullptr;     // cell side, on the first mse_test of a mask: the matrix value at every listed (gene, cell), 0 where A has none
    bool val_ok = false, val_refused = false;
    size_t cap_ptr = 0, cap_idx = 0, cap_val = 0;
    int64_t ncol = 0, total = 0;
    int32_t nrow = 0;
    uint64_t seed = 0, inv = 0;
    int mask_t = -1;           // -1: nothing built
    int64_t col_off = 0, row_off = 0;
    bool refused = false;      // too large for the memory budget under this key: the hashing kernel runs
};

struct sgl_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    DevCSC A, At;  // A: genes x local cells; At: local cells x genes
    DevTiled TA, TAt;  // re-blocked streams of A / At for the current k (k <= 64)
    bool use_tiled = false;
    int64_t cell_offset = 0, ncells_total = 0;
    int64_t* col_nnz_A = nullptr;   // n_local: nnz per cell (skip rule of predict, l.340)
    int64_t* col_nnz_At = nullptr;  // m: nnz per gene of THIS shard
    int64_t* col_nnz_At_global = nullptr;  // m: nnz per gene over all shards (valid while gene_nnz_global)
    bool gene_nnz_global = false;

    int k = 0;
    double *W = nullptr, *Wprev = nullptr, *H = nullptr, *d = nullptr;
    double* B = nullptr;    // k x ncol_local right-hand sides of the H-update
    double* red = nullptr;  // [k*m right-hand sides of the W-update | k*k Gram of H | k row sums]
    double* G = nullptr;    // k x k Gram (+1e-15 diagonal)
    double* Gpad = nullptr; // KP x KP zero-padded copy for the lane NNLS kernel
    NnlsScratch nnls_scr;   // lists / per-column state of the multi-pass lane NNLS
    bool solve_empty = false;  // dense front-end (src/singlet.cpp:370-381): no empty-column skip
    // dense front-end (dense_frontend.hip): the matrix as R holds it (nrow x ncol, column-major) next to its CSC image;
    // dense_gemm: more than half of it is non-zero -> the right-hand sides of predict are GEMMs (rocBLAS)
    double* Adense = nullptr;
    bool dense_gemm = false;
    void* rocblas = nullptr;
    double* link_h = nullptr;  // c_linked_nmf: link_rows x ncol / x nrow multipliers of the right-hand sides
    double* link_w = nullptr;
    int link_h_rows = 0, link_w_rows = 0;
    DevMaskList ML[2];         // masked path: the drawn rows of every column, per orientation (kernels_mask.hip); live with the entry streams
    int64_t ml_builds[2] = {0, 0};        // times the lists of an orientation were hashed out on this context (sgl_layout_builds)
    DevMaskList MLkeep[2][SGL_ML_KEEP];   // the lists of the masks used before this one, most recent first (sgl_mask_list_select)
    double* Gcols = nullptr;   // masked path: per-column Grams of one chunk of columns (gcols_chunk * k * k)
    int64_t gcols_chunk = 0;
    double* Wd = nullptr;      // mse_test: W' = W^T diag(d) as k x m
    double* Sbuf = nullptr;    // team masked W-update: per-gene downdate partials, (gene blocks x team size) x k x k
    double* Stri = nullptr;    // ... their lower triangles, what the reduce-scatter moves: (gene blocks x team size) x k (k + 1) / 2
    struct sgl_team* team = nullptr;   // native collective (multi.hip): the team this context is a rank of
    int team_rank = 0;
    double* ws = nullptr;   // partial-reduction workspace
    size_t ws_bytes = 0;
    double* scalars = nullptr;       // device scratch for cor / mse results
    unsigned long long* sweep_counters = nullptr;   // device: [0] H sweeps, [1] W sweeps
    int64_t sweeps_acc[4] = {0, 0, 0, 0};
    int64_t wave_sweeps_acc[2] = {0, 0};
    double* pinned = nullptr;        // host pinned scratch (8 doubles)

    sgl_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;

    bool timing = false;
    std::vector<PhaseEvent> pending;
    std::vector<hipEvent_t> event_pool;
    double phase_ms[SGL_PH_COUNT] = {0};
    int64_t phase_calls[SGL_PH_COUNT] = {0};
};

// native team (multi.hip): ranks of a cell-sharded fit that exchange through RCCL (or, for ranks that
// share one device inside one process, through a HIP kernel)
int sgl_team_size(const sgl_ctx* c);   // 1 without a team
// internals of singlet_hip.hip used by multi.hip
int sgl_nnls_shared(sgl_ctx* c, const double* G, double* B, double* X, const int64_t* col_nnz, int64_t ncols,
                    double L1, double L2, unsigned long long* counter, bool h_side = false);
int sgl_scale_w_enqueue(sgl_ctx* c);            // scale(w, d); cor(w, w_prev) -> device scalar
int sgl_scale_w_fetch(sgl_ctx* c, double* tol); // copy it out (synchronises the stream)
int sgl_fetch_sweeps(sgl_ctx* c);
// masked path pieces (singlet_hip.hip) used by the team's sharded c_ard_nmf (multi.hip)
int sgl_mask_workspace(sgl_ctx* c);
int sgl_predict_mask_dev(sgl_ctx* c, const DevCSC& M, const int64_t* col_nnz, const double* F, double* X, double* Bbuf,
                         uint64_t seed, uint64_t inv_density, double L1, double L2, int mask_t, int rhs_phase, int nnls_phase,
                         unsigned long long* counter);
int sgl_mse_test_enqueue(sgl_ctx* c, uint64_t seed, uint64_t inv_density);   // local loss sum -> c->scalars[1]
int sgl_ard_run_team(sgl_ctx* c, double tol, int32_t maxit, double L1, double L2, uint64_t seed, uint64_t inv_density,
                     double overfit_threshold, int32_t trace_test_mse, double* test_mse, int32_t* iter, double* tol_out,
                     double* score_overfit, int32_t* n_trace, int32_t* n_iter, const sgl_callbacks* cb);
int sgl_c_ard_nmf_multi(int ndev, const double* Ax, const int32_t* Ai, const int32_t* Ap, int32_t nrow, int32_t ncol, double tol,
                        uint16_t maxit, double L1, double L2, const double* w_init, int32_t k, uint64_t seed, uint64_t inv_density,
                        double overfit_threshold, uint16_t trace_test_mse, double* w_out, double* d_out, double* h_out,
                        double* test_mse, int32_t* iter, double* tol_out, double* score_overfit, int32_t* n_trace,
                        const sgl_callbacks* cb);
#define SGL_MASK_MAX_K SGL_MAX_K   // the masked (ARD) path: MFMA Gram downdates up to k = 128, the generic VALU kernel above
void sgl_team_detach(sgl_ctx* c);               // called by sgl_destroy
int sgl_c_nmf_multi(int ndev, const double* Ax, const int32_t* Ai, const int32_t* Ap, int32_t nrow, int32_t ncol, double tol,
                    uint16_t maxit, double L1_w, double L1_h, double L2_w, double L2_h, const double* w_init, int32_t k,
                    double* w_out, double* d_out, double* h_out, int32_t* n_iter, double* tol_trace, const sgl_callbacks* cb);

// RAII-free phase timer helpers (driver side).
int sgl_phase_begin(sgl_ctx* c, int phase, PhaseEvent* pe);
int sgl_phase_end(sgl_ctx* c, PhaseEvent* pe);
int sgl_ws_reserve(sgl_ctx* c, size_t bytes);

// ---- kernel launchers (each enqueues on `s`, returns SGL_* code) ----------
// hash / synthetic generator / mask
int k_rand(hipStream_t s, uint64_t state, const uint64_t* i, const uint64_t* j, int64_t n, uint64_t* out);
int k_mask(hipStream_t s, uint64_t state, uint64_t inv_density, int64_t cell0, int32_t ncells, int32_t ngenes, uint8_t* out);
// counts per column of the synthetic matrix.  transposed = 0: column = cell
// (rows = genes); 1: column = gene (rows = local cells).
int k_synth_count(hipStream_t s, uint64_t S, uint64_t inv_density, int transposed, int64_t cell_offset,
                  int32_t ncells, int32_t ngenes, int64_t* counts, const double* skew_dev = nullptr);
int k_synth_fill(hipStream_t s, uint64_t S, uint64_t inv_density, const double* levels16_dev, int transposed,
                 int64_t cell_offset, int32_t ncells, int32_t ngenes, const int64_t* p, int32_t* idx, double* x,
                 const double* skew_dev = nullptr);
int k_synth_winit(hipStream_t s, uint64_t S, int k, int32_t ngenes, double* W);
int k_exclusive_scan(sgl_ctx* c, const int64_t* in, int64_t* out, int64_t n);  // out has n+1 entries
int k_scan_total(hipStream_t s, const int64_t* in, int64_t* out, int64_t n);
int k_col_counts(hipStream_t s, const int64_t* p, int64_t ncol, int64_t* counts);
int k_validate_csc(hipStream_t s, const int32_t* idx, const int64_t* p, int64_t ncol, int32_t nrow, int* flag_dev);
int k_all_finite(hipStream_t s, const double* x, int64_t n, int* flag_dev);   // flag |= 4 on NaN / Inf (after k_validate_csc)
int k_widen_p(hipStream_t s, const int32_t* p32, int64_t n1, int64_t* p64);
int k_build_segments(hipStream_t s, const DevCSC& M);

// dense helpers
int k_gram(sgl_ctx* c, const double* F, int k, int64_t cols, double* G, double diag_add);
int k_gram_add_diag(hipStream_t s, double* G, int k, double v);
int k_rowsum(sgl_ctx* c, const double* F, int k, int64_t cols, double* d_out, int add_eps = 0);  // add_eps: + 1e-15 in the final stage
bool k_rowsum_can_add_eps(int k, int64_t cols);
int k_scale_apply(hipStream_t s, double* F, int k, int64_t cols, double* d, int add_eps);
int k_cor(sgl_ctx* c, const double* x, const double* y, int64_t n, double* out_dev);
int k_pad_gram(hipStream_t s, const double* G, int k, int KP, int GS, double* Gpad);
int k_gram_rescale(hipStream_t s, double* G, int k, const double* d, double diag_add);  // G[i,j] = G[i,j] / d_i / d_j (+ diag_add)
int k_i64_to_f64(hipStream_t s, const int64_t* in, double* out, int64_t n);
int k_f64_to_i64(hipStream_t s, const double* in, int64_t* out, int64_t n);
int k_transpose_dense(hipStream_t s, const double* in, int rows, int cols, double* out);

// sparse accumulate: B[:, c] (+)= sum_{nz in tile t of column c} x * F[:, row]
int k_acc(hipStream_t s, const DevCSC& M, const double* F, int k, double* B,
          uint64_t mask_seed, uint64_t inv_density, int mask_mode, int64_t mask_col_offset, int64_t mask_row_offset);

// LDS-tiled accumulate
void sgl_tiled_free(DevTiled& S);
int sgl_tiled_build(sgl_ctx* c, const DevCSC& M, int k, DevTiled& S);
int k_acc_tiled(hipStream_t s, const DevTiled& S, const double* F, int ldf, double* B, int ldb, int kf, const double* xvals = nullptr);
int k_acc_tiled_all(hipStream_t s, const DevTiled& S, const double* F, double* B, int k, const double* xvals = nullptr);
int sgl_tiled_mask_values(sgl_ctx* c, const DevCSC& M, DevTiled& S, uint64_t seed, uint64_t inv_density, int mask_t,
                          int64_t col_off, int64_t row_off);
// masked right-hand sides of predict_mask for one orientation (0: A, 1: At), through the tiled kernel when the fit has one
int sgl_masked_rhs(sgl_ctx* c, int orientation, const double* F, double* Bbuf, uint64_t seed, uint64_t inv_density);
int tiled_part_size(int k);

// dense front-end (dense_frontend.hip)
int k_dense_count(hipStream_t s, const double* A, int32_t nrow, int64_t ncol, int64_t* counts);
int k_dense_fill(hipStream_t s, const double* A, int32_t nrow, int64_t ncol, const int64_t* p, int32_t* idx, double* x);
int k_dense_rhs(sgl_ctx* c, int which, const double* F, int k, double* B);
void sgl_dense_release(sgl_ctx* c);
bool sgl_dense_gemm_available();

// input staging (kernels_prep.hip)
int k_colsum(hipStream_t s, const DevCSC& M, double* sums);
int k_cell_factor(hipStream_t s, DevCSC& A, DevCSC& At, const double* f, int mode, double scale);
int k_link_mul(hipStream_t s, double* B, const double* L, int k, int link_rows, int64_t ncols);

// NNLS
#define SGL_NNLS_MAX_PASSES 10
// below this many columns the GPU is not full anyway: one pass (env SGL_NNLS_REPACK_MIN_COLS overrides, tests)
int64_t nnls_repack_min_cols();
int nnls_gram_stride(int KP);  // row stride of the padded Gram the lane kernel expects
int nnls_lane_kp(int k);
int nnls_scratch_alloc(NnlsScratch& sc, int64_t cap, int k_for_xt);
void nnls_scratch_free(NnlsScratch& sc);
// B is destroyed (and used as the spill space of b between passes).  scr == nullptr: one pass.
int k_nnls_lane(hipStream_t s, const double* Gpad, int KP, double* B, double* X, const int64_t* col_nnz,
                int k, int64_t ncols, double L1, double L2, unsigned long long* sweep_counter, const NnlsScratch* scr,
                bool pack_by_sweeps = false);
int nnls_pack_alloc(NnlsScratch& sc, int64_t ncols);
// ranks 129 - 256, shared Gram, all columns of the shard: four lanes per column, waves packed by the previous solve's sweep counts
int k_nnls_quarter_packed(hipStream_t s, const double* G, const double* B, double* X, const int64_t* col_nnz, int k, int64_t ncols, double L1,
                          double L2, unsigned long long* sweep_counter, const NnlsScratch* scr);
// few columns against a shared Gram (k <= 64): four columns per wave, the Gram staged in LDS; B is read only
int k_nnls_quad_shared(hipStream_t s, const double* G, const double* B, double* X, const int64_t* col_nnz, int k, int64_t ncols,
                       double L1, double L2, unsigned long long* sweep_counter);
// THE DISPATCH of every solve that is not the lane / two-lane / shared-quad kernel (until round 6 named k_nnls_wave, after its
// last resort): per-column Grams (gstride = k * k: the masked path, one GPU AND the gene blocks of a team, multi.hip) run four
// columns per wave -- nnls_quad_kernel (triangles in LDS, k < 40 / 47) or nnls_quad_global_kernel (k <= 112 / 128) -- and only
// above that, or with a shared Gram above k = 128 (gstride = 0), the wave-per-column kernel
int k_nnls_percol(hipStream_t s, const double* G, int64_t gstride, const double* B, double* X, const int64_t* col_nnz,
                int k, int64_t ncols, double L1, double L2, unsigned long long* sweep_counter);

// masked path
int k_mask_gram_cols(hipStream_t s, int64_t col0, int64_t ncols, int32_t nrow, const int64_t* col_nnz,
                     const double* F, const double* G, int k, uint64_t seed, uint64_t inv_density, int mask_t,
                     int64_t col_offset, int64_t row_offset, double* Gcols, const DevMaskList* L = nullptr);
int sgl_mask_list_build(sgl_ctx* c, DevMaskList& L, int64_t ncol, int32_t nrow, uint64_t seed, uint64_t inv_density, int mask_t,
                        int64_t col_offset, int64_t row_offset);
void sgl_mask_list_free(DevMaskList& L);
// the lists of orientation `which` (0: cells, 1: genes) under this key in c->ML[which]: the current ones, ones kept from an
// earlier fit, or built now -- the lists they replace are kept (R's rank search refits one matrix with the seeds
// seed + 1 .. seed + n_replicates over and over, R/ard_nmf.R:95-160; the mask does not depend on the rank)
int sgl_mask_list_select(sgl_ctx* c, int which, int64_t ncol, int32_t nrow, uint64_t seed, uint64_t inv_density, int mask_t,
                         int64_t col_offset, int64_t row_offset);
void sgl_mask_lists_free_all(sgl_ctx* c);
bool sgl_mask_lists_release_kept(sgl_ctx* c);   // memory pressure: drop the kept (not the current) masks; true if anything was freed
int k_mask_gram_finalize(hipStream_t s, const double* G, const double* S, int k, int64_t ncols, double* out);
int k_tri_pack(hipStream_t s, const double* S, int k, int64_t ncols, double* tri);   // lower triangles of ncols symmetric k x k blocks
int k_mask_gram_finalize_tri(hipStream_t s, const double* G, const double* tri, int k, int64_t ncols, double* out);
int k_mse_test(sgl_ctx* c, const double* Wd, const double* H, int k, uint64_t seed, uint64_t inv_density,
               double* out_dev);
int k_wd(hipStream_t s, const double* W, const double* d, int k, int64_t cols, double* Wd);

// ---- device-side hash (rng::rand, src/singlet.cpp:30-64) ------------------
__device__ __forceinline__ uint64_t sgl_rand2(uint64_t state, uint64_t i, uint64_t j) {
    i ^= i << 19;
    i ^= i >> 7;
    i ^= i << 36;
    uint64_t x = state + i;
    x ^= x << 38;
    x ^= x >> 13;
    x ^= x << 23;
    j ^= j >> 7;
    j ^= j << 23;
    j ^= j >> 8;
    x += j;
    x ^= x >> 7;
    x ^= x << 53;
    x ^= x >> 4;
    return x;
}
// the i-only half of the hash, hoistable out of loops over j
__device__ __forceinline__ uint64_t sgl_rand_i(uint64_t state, uint64_t i) {
    i ^= i << 19;
    i ^= i >> 7;
    i ^= i << 36;
    uint64_t x = state + i;
    x ^= x << 38;
    x ^= x >> 13;
    x ^= x << 23;
    return x;
}
__device__ __forceinline__ uint64_t sgl_rand_j(uint64_t xi, uint64_t j) {
    j ^= j >> 7;
    j ^= j << 23;
    j ^= j >> 8;
    uint64_t x = xi + j;
    x ^= x >> 7;
    x ^= x << 53;
    x ^= x >> 4;
    return x;
}
// `x % d == 0` for a kernel-uniform divisor without a 64-bit division per test (hipcc expands a runtime u64
// modulo into a long software routine, the bulk of the hashing cost of the masked path).  With d = 2^s * o
// (o odd):  d | x  <=>  the low s bits of x are zero  and  o | (x >> s);  and for odd o (Granlund-Montgomery)
// o | y  <=>  y * o^-1 (mod 2^64) <= floor((2^64 - 1) / o).  One 64-bit low multiply and a compare instead of
// round 1's multiply-high + multiply-low + two corrections.  Exact for every x and d >= 1 (d = 0 is rejected
// on the host); tests/test_gpu_ops.py::test_mask_bit_exact holds it against the oracle's `%`.
struct SglDiv {
    uint64_t d, low_mask, inv, lim;
    int shift;
};
static inline SglDiv sgl_div_make(uint64_t d) {
    SglDiv v{d, 0ull, 1ull, ~0ull, 0};
    if (d <= 1) return v;
    uint64_t o = d;
    while ((o & 1ull) == 0) { o >>= 1; ++v.shift; }
    v.low_mask = (1ull << v.shift) - 1ull;
    uint64_t inv = o;                       // Newton: doubles the correct low bits each step (3 -> 6 -> ... -> 96)
    for (int q = 0; q < 6; ++q) inv *= 2ull - o * inv;
    v.inv = inv;
    v.lim = ~0ull / o;
    return v;
}
__device__ __forceinline__ bool sgl_divides(uint64_t x, SglDiv dv) {
    if (dv.d <= 1) return true;
    if ((x & dv.low_mask) != 0ull) return false;
    return (x >> dv.shift) * dv.inv <= dv.lim;
}
__device__ __forceinline__ bool sgl_draw(uint64_t state, uint64_t i, uint64_t j, SglDiv inv_density) {
    return sgl_divides(sgl_rand2(state, i, j), inv_density);
}
