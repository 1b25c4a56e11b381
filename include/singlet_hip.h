/*
 * singlet_hip.h -- C ABI of libsinglet_hip.so, the MI355X (gfx950) engine for
 * singlet's alternating-least-squares hot path.
 *
 * This is the drop-in boundary.  The reference's R wrappers c_nmf / c_ard_nmf /
 * c_project_model (R/RcppExports.R:24-30, 78-80) reach C++ through the Rcpp
 * glue _singlet_c_nmf / _singlet_c_ard_nmf / _singlet_c_project_model
 * (src/RcppExports.cpp:98-116, 284-304, 444-447).  A maintainer replaces the
 * bodies of those three glue functions by calls to section 1 below (stub in
 * INTEGRATION.md); everything R-side stays unchanged.
 *
 * Conventions
 *  - plain C types only; no exceptions cross the ABI.  Every function returns
 *    0 on success or a negative SGL_E* code; sgl_last_error() gives the text
 *    (the R shim turns it into Rf_error, as END_RCPP does for exceptions,
 *    src/RcppExports.cpp:115).
 *  - sparse matrices are dgCMatrix slots (inst/include/singlet.h:36-44):
 *    x double[nnz], i int32[nnz] (row index, ascending within a column),
 *    p int32[ncol+1], Dim = (nrow, ncol).
 *  - dense matrices are column-major doubles exactly as R / Eigen hold them:
 *    w is k x nrow(A), h is k x ncol(A).
 *  - input pointers are never written through nor retained after return;
 *    outputs go to caller-allocated buffers.
 *  - all arithmetic on the path is FP64 on the GPU; there is no CPU fallback:
 *    without a usable gfx950 device every entry point fails with SGL_ENODEV.
 *  - ranks: every entry point takes 1 <= k <= 1024 (SGL_EINVAL above, before anything is uploaded; the
 *    reference's nnls / predict_mask have no limit, src/singlet.cpp:229-250, 436-466).  The tuned kernels cover
 *    k <= 128 (LDS-tiled accumulate, MFMA Grams and Gram downdates, lane NNLS); the plain fit runs ranks 129 - 256 on the
 *    same entry streams, with the Gram on the matrix cores and four lanes per column in the solve (round 6: an iteration
 *    at k = 130 costs 1.3 x one at k = 128); above 256, and for the masked Gram downdate above 128, generic kernels run
 *    (wave-per-column NNLS, VALU Grams in several launches): correct, several times slower.
 *  - device memory: blocks of 64 MB and more that a call frees are kept for the next call (sgl_pool_info,
 *    sgl_cache_release; SGL_POOL=0 switches it off).
 *  - several GPUs: section 2b; the one-shot sgl_c_nmf / sgl_c_ard_nmf honour SINGLET_NGPU.
 */
#ifndef SINGLET_HIP_H
#define SINGLET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGL_API __attribute__((visibility("default")))

#define SGL_OK 0
#define SGL_EINVAL (-1)   /* bad argument */
#define SGL_ENODEV (-2)   /* no HIP device / wrong architecture */
#define SGL_EHIP (-3)     /* a HIP runtime call failed */
#define SGL_ENOMEM (-4)   /* device or host allocation failed */
#define SGL_EINTR (-5)    /* the poll callback asked to stop (Rcpp::checkUserInterrupt) */
#define SGL_ESTATE (-6)   /* call out of order on a context */
#define SGL_ECOMM (-7)    /* the all-reduce callback failed, or RCCL is missing / returned an error */

SGL_API const char* sgl_last_error(void);
/* ABI version of this header; bumped on any signature change. */
SGL_API int sgl_abi_version(void);   /* 2: native multi-GPU section, sgl_nmf_iterate, chunk-list / dense ARD entry points */
/* Number of usable gfx950 devices (0 if none); never fails. */
SGL_API int sgl_device_count(void);

/* Callbacks, all invoked on the calling thread only (as Rprintf and
 * Rcpp::checkUserInterrupt are in the reference, src/singlet.cpp:643-663). */
typedef struct sgl_callbacks {
    void* user;
    /* per-iteration trace line: iter is 1-based as printed by the reference
     * (src/singlet.cpp:661-662, 1115-1127); overfit is NaN where the reference
     * prints "-" or has no such column. */
    void (*log)(void* user, int iter, double tol, double overfit);
    /* return non-zero to abort; polled at the two points where the reference
     * calls Rcpp::checkUserInterrupt() (src/singlet.cpp:652, 663). */
    int (*poll)(void* user);
} sgl_callbacks;

/* ------------------------------------------------------------------------
 * 1. One-shot entry points: exactly what the Rcpp glue binds.
 * ---------------------------------------------------------------------- */

/* c_nmf (src/singlet.cpp:669-672 -> c_nmf_base :638-666).
 * Replaces _singlet_c_nmf (src/RcppExports.cpp:98-116).
 * A is nrow x ncol (genes x cells); At is its transpose (ncol x nrow) as
 * R/run_nmf.R:40 builds it; Atx/Ati/Atp may all be NULL, then the transpose is
 * built on the device.  `threads` and `verbose` are accepted for signature
 * parity (threads is meaningless on the GPU; verbose output goes through
 * cb->log).  w_init: k x nrow.  Outputs: w_out k x nrow, d_out k, h_out k x ncol;
 * *n_iter = iterations run; tol_trace (optional, maxit doubles) = tol per
 * iteration. */
SGL_API int sgl_c_nmf(const double* Ax, const int32_t* Ai, const int32_t* Ap,
              const double* Atx, const int32_t* Ati, const int32_t* Atp,
              int32_t nrow, int32_t ncol,
              double tol, uint16_t maxit, int verbose,
              double L1_w, double L1_h, double L2_w, double L2_h, uint16_t threads,
              const double* w_init, int32_t k,
              double* w_out, double* d_out, double* h_out,
              int32_t* n_iter, double* tol_trace, const sgl_callbacks* cb);

/* Opt-in resident matrix between one-shot calls: with SINGLET_HIP_CACHE=1 in the environment sgl_c_nmf and
 * sgl_c_ard_nmf keep the context of their last call and skip upload / validation / transpose when the next call
 * passes the same host slots (same pointers, shape, non-zero count and a sampled fingerprint of x, i, p) -- the
 * unchanged R drivers' rank sweep (R/ard_nmf.R:95-160, R/cross_validate_nmf.R:69-97) then runs on a resident A.
 * The pointers are compared, never dereferenced later.  Unset (default): every call uploads.
 * sgl_cache_release frees the kept context (and its device memory); call it before unloading the library. */
SGL_API int sgl_cache_release(void);

/* c_ard_nmf (src/singlet.cpp:1155-1159 -> c_ard_nmf_base :1090-1152).
 * Replaces _singlet_c_ard_nmf (src/RcppExports.cpp:284-304).
 * Trace arrays (test_mse, iter, tol, score_overfit) must hold maxit + 1
 * entries; *n_trace receives their used length (the reference returns them as
 * R vectors, src/singlet.cpp:1144-1151).  Ranks as in the header's preamble (k <= 1024). */
SGL_API int sgl_c_ard_nmf(const double* Ax, const int32_t* Ai, const int32_t* Ap,
                  const double* Atx, const int32_t* Ati, const int32_t* Atp,
                  int32_t nrow, int32_t ncol,
                  double tol, uint16_t maxit, int verbose,
                  double L1, double L2, uint16_t threads,
                  const double* w_init, int32_t k,
                  uint64_t seed, uint64_t inv_density, double overfit_threshold, uint16_t trace_test_mse,
                  double* w_out, double* d_out, double* h_out,
                  double* test_mse, int32_t* iter, double* tol_out, double* score_overfit, int32_t* n_trace,
                  const sgl_callbacks* cb);

/* c_nmf_dense (src/singlet.cpp:1052-1054; dense predict :370-381), the branch
 * R/run_nmf.R:57 takes for a dense matrix.  Replaces _singlet_c_nmf_dense
 * (11 args; At is not needed).  A: nrow x ncol column-major doubles.  Unlike the
 * sparse path, all-zero columns are solved too (the dense predict has no
 * empty-column skip). */
SGL_API int sgl_c_nmf_dense(const double* A, int32_t nrow, int32_t ncol,
                    double tol, uint16_t maxit, int verbose,
                    double L1_w, double L1_h, double L2_w, double L2_h, uint16_t threads,
                    const double* w_init, int32_t k,
                    double* w_out, double* d_out, double* h_out,
                    int32_t* n_iter, double* tol_trace, const sgl_callbacks* cb);

/* c_ard_nmf_dense (src/singlet.cpp:1357-1361; dense predict_mask :506-533, mse_test :608-632), the branch
 * R/ard_nmf.R:109 and R/cross_validate_nmf.R:82 take for a dense matrix.  Replaces _singlet_c_ard_nmf_dense
 * (13 args; At is not needed).  Every column is solved, as in sgl_c_nmf_dense. */
SGL_API int sgl_c_ard_nmf_dense(const double* A, int32_t nrow, int32_t ncol,
                        double tol, uint16_t maxit, int verbose,
                        double L1, double L2, uint16_t threads,
                        const double* w_init, int32_t k,
                        uint64_t seed, uint64_t inv_density, double overfit_threshold, uint16_t trace_test_mse,
                        double* w_out, double* d_out, double* h_out,
                        double* test_mse, int32_t* iter, double* tol_out, double* score_overfit, int32_t* n_trace,
                        const sgl_callbacks* cb);

/* c_nmf_sparse_list (src/singlet.cpp:715-743) and c_ard_nmf_sparse_list (:1162-1234): A as a LIST of column
 * chunks (each nrow x chunk_ncol[q], dgCMatrix slots), as R/ard_nmf.R:114,181 passes it; the reference walks
 * the chunks with a running column offset (:384-402, :485, :590).  Replace _singlet_c_nmf_sparse_list (9 args)
 * and _singlet_c_ard_nmf_sparse_list (13 args).  The chunks are joined on the device into one resident matrix
 * (64-bit column pointers: the total may exceed one dgCMatrix's 2^31 - 1 non-zeros).  At_: the list of column
 * chunks of t(A) (each ncol_total x t_chunk_ncol[q]); n_t_chunks = 0 builds the transpose on the device. */
SGL_API int sgl_c_nmf_sparse_list(int32_t n_chunks, const double* const* Ax, const int32_t* const* Ai, const int32_t* const* Ap,
                          const int32_t* chunk_ncol,
                          int32_t n_t_chunks, const double* const* Atx, const int32_t* const* Ati, const int32_t* const* Atp,
                          const int32_t* t_chunk_ncol,
                          int32_t nrow,
                          double tol, uint16_t maxit, int verbose, double L1, double L2, uint16_t threads,
                          const double* w_init, int32_t k,
                          double* w_out, double* d_out, double* h_out,
                          int32_t* n_iter, double* tol_trace, const sgl_callbacks* cb);
SGL_API int sgl_c_ard_nmf_sparse_list(int32_t n_chunks, const double* const* Ax, const int32_t* const* Ai, const int32_t* const* Ap,
                              const int32_t* chunk_ncol,
                              int32_t n_t_chunks, const double* const* Atx, const int32_t* const* Ati, const int32_t* const* Atp,
                              const int32_t* t_chunk_ncol,
                              int32_t nrow,
                              double tol, uint16_t maxit, int verbose, double L1, double L2, uint16_t threads,
                              const double* w_init, int32_t k,
                              uint64_t seed, uint64_t inv_density, double overfit_threshold, uint16_t trace_test_mse,
                              double* w_out, double* d_out, double* h_out,
                              double* test_mse, int32_t* iter, double* tol_out, double* score_overfit, int32_t* n_trace,
                              const sgl_callbacks* cb);

/* c_linked_nmf (src/singlet.cpp:1059-1086; predict_link :416-433), the linked
 * NMF behind R/RunLNMF.R:60.  Replaces _singlet_c_linked_nmf (11 args).  link_h
 * (link_h_rows x link_h_cols, column-major) multiplies the first link_h_rows
 * entries of every cell's right-hand side before its NNLS solve; it is applied
 * iff link_h_cols == ncol (l.1064).  link_w likewise for genes, iff
 * link_w_cols == nrow (l.1065).  Either may be NULL. */
SGL_API int sgl_c_linked_nmf(const double* Ax, const int32_t* Ai, const int32_t* Ap,
                     const double* Atx, const int32_t* Ati, const int32_t* Atp,
                     int32_t nrow, int32_t ncol,
                     double tol, uint16_t maxit, int verbose,
                     double L1, double L2, uint16_t threads,
                     const double* w_init, int32_t k,
                     const double* link_h, int32_t link_h_rows, int32_t link_h_cols,
                     const double* link_w, int32_t link_w_rows, int32_t link_w_cols,
                     double* w_out, double* d_out, double* h_out,
                     int32_t* n_iter, double* tol_trace, const sgl_callbacks* cb);

/* c_project_model (src/singlet.cpp:405-413).
 * Replaces _singlet_c_project_model (src/RcppExports.cpp:444-447 region).
 * w is w_rows x w_cols column-major; if w_rows == nrow it is transposed first
 * (l.406).  k = the factor dimension after that.  h_out: k x ncol, d_out: k. */
SGL_API int sgl_c_project_model(const double* Ax, const int32_t* Ai, const int32_t* Ap,
                        int32_t nrow, int32_t ncol,
                        const double* w, int32_t w_rows, int32_t w_cols,
                        double L1, double L2, uint16_t threads,
                        double* h_out, double* d_out);

/* Rcpp_predict (src/singlet.cpp:350-367): the projection without the two
 * scale() calls.  Replaces _singlet_Rcpp_predict (src/RcppExports.cpp, 5 args,
 * registered at :462).  w is transposed iff w_rows == nrow && w_cols != nrow
 * (l.351 -- not the same rule as c_project_model's).  h_out: k x ncol. */
SGL_API int sgl_rcpp_predict(const double* Ax, const int32_t* Ai, const int32_t* Ap,
                             int32_t nrow, int32_t ncol,
                             const double* w, int32_t w_rows, int32_t w_cols,
                             double L1, double L2, uint16_t threads, double* h_out);

/* ------------------------------------------------------------------------
 * 2. Context API: the same path with the matrix kept resident in HBM, for
 *    rank sweeps (R/ard_nmf.R:95-160 calls c_ard_nmf many times on one A),
 *    for cell-sharded multi-GPU runs and for the benchmark.  One context =
 *    one device = one shard of cells.
 * ---------------------------------------------------------------------- */
typedef struct sgl_ctx sgl_ctx;

SGL_API int sgl_create(int device, sgl_ctx** out);
SGL_API int sgl_destroy(sgl_ctx* ctx);
/* Launch all work of this context on `hip_stream` (a hipStream_t; NULL = the
 * context's own stream).  Lets a host that owns a stream (torch's current
 * stream) order collectives against the kernels without host syncs. */
SGL_API int sgl_set_stream(sgl_ctx* ctx, void* hip_stream);

/* Upload a shard: columns [cell_offset, cell_offset + ncol) of a genes x
 * ncells_total matrix.  At describes the same shard transposed (ncol x nrow,
 * row indices local to the shard); pass NULLs to have it built on the device.
 * Replaces the Rcpp::SparseMatrix views (inst/include/singlet.h:108-127). */
SGL_API int sgl_upload_csc(sgl_ctx* ctx, const double* Ax, const int32_t* Ai, const int32_t* Ap,
                   const double* Atx, const int32_t* Ati, const int32_t* Atp,
                   int32_t nrow, int32_t ncol, int64_t cell_offset, int64_t ncells_total);

/* The same from a list of column chunks (sgl_c_nmf_sparse_list above): the chunks become ONE resident shard. */
SGL_API int sgl_upload_csc_list(sgl_ctx* ctx, int32_t n_chunks, const double* const* Ax, const int32_t* const* Ai,
                        const int32_t* const* Ap, const int32_t* chunk_ncol,
                        int32_t n_t_chunks, const double* const* Atx, const int32_t* const* Ati, const int32_t* const* Atp,
                        const int32_t* t_chunk_ncol,
                        int32_t nrow, int64_t cell_offset, int64_t ncells_total);

/* A dense matrix (nrow x ncol, column-major doubles, as R / Eigen hold it; what c_nmf_dense and c_ard_nmf_dense take,
 * src/singlet.cpp:1052-1054, 1357-1361): resident as its CSC image (zeros dropped, both orientations, built on the
 * device) and -- when more than half of its entries are non-zero -- as the dense copy itself, on which the plain fit then
 * forms the right-hand sides of predict (`w * A.col(i)`, src/singlet.cpp:377) as FP64 GEMMs (rocBLAS, bound at run time).
 * A fit on it solves every column like the reference's dense predict (no empty-column skip) when the caller asks for it
 * (sgl_c_nmf_dense does). */
SGL_API int sgl_upload_dense(sgl_ctx* ctx, const double* A, int32_t nrow, int32_t ncol);

/* Generate the synthetic benchmark shard on the device:
 * entry (gene g, cell c) non-zero iff rand_S(c,g) % inv_density == 0, value
 * levels16[(rand_{S+1}(c,g) >> 11) % 16].  Both orientations are produced. */
SGL_API int sgl_synth_csc(sgl_ctx* ctx, uint64_t S, uint64_t inv_density, const double* levels16,
                  int32_t ngenes, int64_t cell_offset, int32_t ncells_local, int64_t ncells_total);
/* The same with SKEWED rows and columns (a benchmark matrix nearer to count data than the i.i.d. one; the
 * reference's own fixture pbmc3k has 3 ... 2700 non-zeros per gene): entry (g, c) is non-zero iff
 * u(c, g) < cell_w16[level(c)] * gene_w16[level(g)] / inv_density, u = (rand_S(c, g) >> 11) * 2^-53, the levels
 * 4-bit hashes of the cell / gene index alone.  Both tables NULL = sgl_synth_csc. */
SGL_API int sgl_synth_csc_skewed(sgl_ctx* ctx, uint64_t S, uint64_t inv_density, const double* levels16,
                  int32_t ngenes, int64_t cell_offset, int32_t ncells_local, int64_t ncells_total,
                  const double* cell_w16, const double* gene_w16);

/* Shape / size queries. */
SGL_API int sgl_dims(const sgl_ctx* ctx, int32_t* nrow, int32_t* ncol, int64_t* nnz);
/* Download the resident shard as dgCMatrix slots (tests of sgl_synth_csc and
 * of the device transpose).  which = 0: A, 1: At.  Buffers sized from sgl_dims. */
SGL_API int sgl_download_csc(sgl_ctx* ctx, int which, double* x, int32_t* i, int64_t* p);

/* Input staging on the resident shard, applied to A and its transpose in place
 * (call before sgl_fit_init; a running fit is dropped).
 * sgl_log_normalize: Seurat::LogNormalize as PreprocessData.dgCMatrix applies
 *   it (R/PreprocessData.R:34-39): x <- log1p(x / colSums(A)[cell] * scale_factor).
 * sgl_weight_by_split: weight_by_split (src/singlet.cpp:119-144): split_by[c] in
 *   [0, n_groups) is the group of local cell c; cells of group g != 0 are divided
 *   by (sum of group g) / (sum of group 0).  Group sums are global over shards
 *   (all-reduce hook). */
SGL_API int sgl_log_normalize(sgl_ctx* ctx, double scale_factor);
SGL_API int sgl_weight_by_split(sgl_ctx* ctx, const int32_t* split_by, int32_t n_groups);
/* The one-shot form the Rcpp glue binds (`_singlet_weight_by_split`, src/RcppExports.cpp:17-27, 445; called from
 * R/RunNMF.R:86-93): dgCMatrix slots in, the re-weighted values out.  x_out (nnz doubles) may be the x slot itself:
 * the reference rewrites the values of A in place (src/singlet.cpp:136-141). */
SGL_API int sgl_c_weight_by_split(const double* Ax, const int32_t* Ai, const int32_t* Ap, int32_t nrow, int32_t ncol,
                                  const int32_t* split_by, int32_t n_groups, double* x_out);

/* Start a fit at rank k.  w_init: k x nrow host array, or NULL to fill W on
 * the device with the synthetic init ((rand_{S+2}(f,g) >> 11) + 0.5) * 2^-53.
 * h = 0, d = 1 as in src/singlet.cpp:639-641.
 * May be called again and again on one resident matrix (R's ard_nmf / cross_validate_nmf refit one matrix tens
 * of times, R/ard_nmf.R:95-160): the re-blocked entry streams of the matrix (2 x ~15 bytes per non-zero at
 * k <= 128, + 8 per non-zero once a masked fit has run) are kept between fits -- a fit at an unchanged rank reuses
 * them as they are, another rank rebuilds them in the same allocations -- and are released when the matrix changes
 * (upload, synth, log-normalize, weight_by_split) or the context is destroyed. */
SGL_API int sgl_fit_init(sgl_ctx* ctx, int32_t k, const double* w_init, uint64_t synth_seed);

/* Link matrices of c_linked_nmf for the current fit (after sgl_fit_init; the
 * rules of sgl_c_linked_nmf; link_h columns are the cells of this shard). */
SGL_API int sgl_set_links(sgl_ctx* ctx, const double* link_h, int32_t link_h_rows, int32_t link_h_cols,
                          const double* link_w, int32_t link_w_rows, int32_t link_w_cols);

/* Collective hook for cell-sharded runs.  Called with a device pointer to
 * `count` doubles that must be summed in place over all shards.  Ordering:
 * after sgl_set_stream(ctx, S) the buffer is produced and consumed by kernels
 * on S, so the hook enqueues its collective on S (or a stream ordered against
 * S) and returns without a host sync.  Without sgl_set_stream the context runs
 * on a private stream: the library then synchronises it before the call, and
 * the hook must have completed its writes when it returns.  NULL = one shard. */
typedef int (*sgl_allreduce_fn)(void* user, void* dev_ptr, int64_t count);
SGL_API int sgl_set_allreduce(sgl_ctx* ctx, sgl_allreduce_fn fn, void* user);
/* The hook may be installed or cleared at any time, before or after sgl_fit_init: what depends on it
 * (the per-gene non-zero counts over all shards, which decide the W columns predict() skips,
 * src/singlet.cpp:340) is rebuilt through the new hook at the next W-update. */

/* Step-level operators (what c_nmf_base's loop body is made of).  A sharded
 * host runs them in this order per iteration; sgl_nmf_run does the same
 * internally.
 *   sgl_step_h      predict(A, w, h, L1_h, L2_h)    src/singlet.cpp:650
 *   sgl_step_scale_h  scale(h, d)  (all-reduces k row sums if sharded)  :651
 *   sgl_step_w      predict(At, h, w, L1_w, L2_w)   :654  (all-reduces the
 *                   k x nrow right-hand sides and the k x k Gram if sharded)
 *   sgl_step_scale_w  scale(w, d); tol = cor(w, w_prev)  :655-659
 * sgl_step_begin snapshots w_it = w (:648). */
SGL_API int sgl_step_begin(sgl_ctx* ctx);
SGL_API int sgl_step_h(sgl_ctx* ctx, double L1, double L2);
SGL_API int sgl_step_scale_h(sgl_ctx* ctx);
SGL_API int sgl_step_w(sgl_ctx* ctx, double L1, double L2);
SGL_API int sgl_step_scale_w(sgl_ctx* ctx, double* tol_out);
/* The masked half-iterations c_ard_nmf_base's loop body is made of (src/singlet.cpp:1104, :1106):
 *   sgl_step_h_masked   predict_mask(A, seed, inv_density, w, h, L1, L2, threads, false)    :436-466
 *   sgl_step_w_masked   predict_mask(At, seed, inv_density, h, w, L1, L2, threads, true)
 * on the resident fit of a single shard (the sharded masked loop is sgl_ard_run / sgl_multi_ard_run); the hash sees
 * the shard's global cell index (cell_offset), as in the reference's chunked form (:485).  sgl_ard_run runs
 * sgl_step_begin, sgl_step_h_masked, sgl_step_scale_h, sgl_step_w_masked, sgl_step_scale_w per iteration. */
SGL_API int sgl_step_h_masked(sgl_ctx* ctx, double L1, double L2, uint64_t seed, uint64_t inv_density);
SGL_API int sgl_step_w_masked(sgl_ctx* ctx, double L1, double L2, uint64_t seed, uint64_t inv_density);

/* Whole loops on the resident shard. */
SGL_API int sgl_nmf_run(sgl_ctx* ctx, double tol, int32_t maxit,
                double L1_w, double L1_h, double L2_w, double L2_h,
                int32_t* n_iter, double* tol_trace, const sgl_callbacks* cb);
SGL_API int sgl_ard_run(sgl_ctx* ctx, double tol, int32_t maxit, double L1, double L2,
                uint64_t seed, uint64_t inv_density, double overfit_threshold, int32_t trace_test_mse,
                double* test_mse, int32_t* iter, double* tol_out, double* score_overfit, int32_t* n_trace,
                int32_t* n_iter, const sgl_callbacks* cb);
/* One H-update against a fixed, already scaled W (the body of c_project_model). */
SGL_API int sgl_project_run(sgl_ctx* ctx, double L1, double L2);

/* Results of the current fit: w k x nrow, d k, h k x ncol_local (any may be NULL). */
SGL_API int sgl_get_factors(sgl_ctx* ctx, double* w, double* d, double* h);
/* Overwrite the current factors (warm start / tests).  Any may be NULL. */
SGL_API int sgl_set_factors(sgl_ctx* ctx, const double* w, const double* d, const double* h);

/* One ALS iteration (src/singlet.cpp:648-659) on a context, whatever its exchange: none (one shard),
 * the all-reduce hook (the five steps above) or a native team (section 2b).  *tol = cor(w, w_prev). */
SGL_API int sgl_nmf_iterate(sgl_ctx* ctx, double L1_w, double L1_h, double L2_w, double L2_h, double* tol);

/* ------------------------------------------------------------------------
 * 2b. Native multi-GPU: cells sharded over the GPUs of one node, the exchange done by the library
 *     itself over RCCL / xGMI (no hook, no launcher, nothing for the R side to do).  Per iteration:
 *     ONE grouped collective -- reduce-scatter of the k x genes right-hand sides of the W-update by
 *     gene blocks + all-reduce of [k x k Gram of h | k row sums of h], all taken from the UNSCALED h
 *     (they commute with scale(h, d), src/singlet.cpp:651) -- then every rank solves its block of
 *     genes and the blocks of w are all-gathered.  W, d and tol come out identical on all ranks.
 *     Results equal the one-GPU fit to rounding (the scaling is applied after the sums).
 *     Limits: c_nmf and c_ard_nmf (no links, no dense front-end), k as for one GPU.
 *     RCCL is loaded at run time (librccl.so.1; SGL_RCCL_PATH overrides); SGL_ECOMM if absent.
 * ---------------------------------------------------------------------- */
/* (a) ONE process drives all devices -- the form an R session uses.  sgl_c_nmf itself takes this
 *     path (and sgl_c_ard_nmf its masked counterpart) when the environment variable SINGLET_NGPU
 *     is set to a number > 1.
 *     devices: ndev device ids (NULL: 0 .. ndev-1), all distinct -> RCCL (ncclCommInitAll); all
 *     equal -> the ranks share one device and exchange through a HIP kernel (test configuration). */
typedef struct sgl_multi sgl_multi;
SGL_API int sgl_multi_create(int ndev, const int* devices, sgl_multi** out);
SGL_API int sgl_multi_destroy(sgl_multi* m);
SGL_API int sgl_multi_size(const sgl_multi* m);
/* rank's context, owned by m (for timing / layout queries; do not destroy). */
SGL_API int sgl_multi_ctx(sgl_multi* m, int rank, sgl_ctx** out);
/* Whole matrix in, cells split into contiguous blocks of (nearly) equal non-zero count; the transposed
 * shards are built on the devices. */
SGL_API int sgl_multi_upload_csc(sgl_multi* m, const double* Ax, const int32_t* Ai, const int32_t* Ap, int32_t nrow, int32_t ncol);
SGL_API int sgl_multi_synth_csc(sgl_multi* m, uint64_t S, uint64_t inv_density, const double* levels16, int32_t ngenes,
                                int64_t ncells_total);
SGL_API int sgl_multi_fit_init(sgl_multi* m, int32_t k, const double* w_init, uint64_t synth_seed);
/* c_linked_nmf on the team: the link matrices of the WHOLE matrix (arguments as sgl_set_links); link_h's columns follow
 * their cells to the ranks.  Call after sgl_multi_fit_init.  (One process per GPU: sgl_set_links on the rank's context with
 * the columns of link_h that belong to its cells.) */
SGL_API int sgl_multi_set_links(sgl_multi* m, const double* link_h, int32_t link_h_rows, int32_t link_h_cols,
                                const double* link_w, int32_t link_w_rows, int32_t link_w_cols);
SGL_API int sgl_multi_iterate(sgl_multi* m, double L1_w, double L1_h, double L2_w, double L2_h, double* tol);
SGL_API int sgl_multi_nmf_run(sgl_multi* m, double tol, int32_t maxit, double L1_w, double L1_h, double L2_w, double L2_h,
                              int32_t* n_iter, double* tol_trace, const sgl_callbacks* cb);
/* c_ard_nmf_base on the team (arguments as sgl_ard_run).  The W-update needs, per gene, sums over ALL cells of
 * the right-hand side and of the Gram downdate over the cells masked for that gene: one grouped collective
 * reduce-scatters [k x genes | k x k x genes] by gene blocks (600 MB at k = 50, 30 000 genes -- as a
 * reduce-scatter each rank moves (N-1)/N of it once) and all-reduces the k x k Gram; mse_test adds one
 * all-reduced double per trace.  scale(h, d) keeps the reference's order (k row sums all-reduced first). */
SGL_API int sgl_multi_ard_run(sgl_multi* m, double tol, int32_t maxit, double L1, double L2,
                              uint64_t seed, uint64_t inv_density, double overfit_threshold, int32_t trace_test_mse,
                              double* test_mse, int32_t* iter, double* tol_out, double* score_overfit, int32_t* n_trace,
                              int32_t* n_iter, const sgl_callbacks* cb);
/* w k x nrow, d k, h k x ncol (all cells, in matrix order); any may be NULL. */
SGL_API int sgl_multi_get_factors(sgl_multi* m, double* w, double* d, double* h);
/* (b) one process per GPU (torch.distributed.run, MPI ...): rank 0 makes an id, the host broadcasts
 *     its SGL_COMM_ID_BYTES bytes, every rank joins with its context BEFORE sgl_fit_init.  The
 *     collectives run on the context's stream; sgl_nmf_iterate / sgl_nmf_run then do the team
 *     iteration.  The communicator is destroyed with the context. */
#define SGL_COMM_ID_BYTES 128
SGL_API int sgl_comm_unique_id(void* id);
SGL_API int sgl_comm_init_rank(sgl_ctx* ctx, int nranks, int rank, const void* id);
/* sgl_comm_init_rank is collective (it blocks until all nranks have joined), so a rank that cannot bind RCCL must be
 * found BEFORE anyone calls it: every rank calls sgl_comm_available (no communication; 0 or SGL_ECOMM; path_out, may
 * be NULL, receives the library name that was opened -- SGL_RCCL_PATH in the environment names it, e.g. the librccl
 * the host process has already loaded), the host agrees on the result, and only then all ranks join. */
SGL_API int sgl_comm_available(char* path_out, int path_len);
/* What the library's own communicator of this context reports: *nranks = ncclCommCount (the team size for ranks that
 * share a device; 1 without a team), *is_rccl = 1 when the exchange runs over RCCL, path_out = the RCCL library bound. */
SGL_API int sgl_comm_info(sgl_ctx* ctx, int32_t* nranks, int32_t* is_rccl, char* path_out, int path_len);
/* The cell split sgl_multi_upload_csc uses: lo[0..n] boundaries of n contiguous blocks of (nearly) equal non-zero
 * count, each at least one cell (p = the dgCMatrix p slot, ncol + 1 entries; ncol >= n).  Host only. */
SGL_API int sgl_split_cells_by_nnz(const int32_t* p, int32_t ncol, int n, int64_t* lo);

/* ------------------------------------------------------------------------
 * 3. Single operators, exposed for the parity tests (each is one kernel
 *    family of the path) and for profiling.
 * ---------------------------------------------------------------------- */
/* rng::rand(i, j) (src/singlet.cpp:47-64) for n (i, j) pairs, on the device. */
SGL_API int sgl_op_rand(sgl_ctx* ctx, uint64_t state, const uint64_t* i, const uint64_t* j, int64_t n, uint64_t* out);
/* draw(cell, gene, inv_density) (src/singlet.cpp:91-95) for the block
 * cells [cell0, cell0+ncells) x genes [0, ngenes): out[c * ngenes + g]. */
SGL_API int sgl_op_mask(sgl_ctx* ctx, uint64_t state, uint64_t inv_density, int64_t cell0, int32_t ncells, int32_t ngenes,
                uint8_t* out);
/* AAt (src/singlet.cpp:200-206): G = F F^T (+1e-15 on the diagonal), F k x cols. */
SGL_API int sgl_op_gram(sgl_ctx* ctx, const double* F, int32_t k, int64_t cols, double* G);
/* The per-column Gram downdate of predict_mask (src/singlet.cpp:458-463): for the columns c in [0, ncols)
 * out[c] (k x k) = G - (AAt(F[:, idx_c]) + 1e-15 I), idx_c = the rows r in [0, nrow) with draw(...) true
 * (mask_t = 0: draw(cell = c + col_offset, gene = r + row_offset); 1: draw(cell = r + row_offset, gene = c + col_offset));
 * G = NULL: the plain sum AAt(F[:, idx_c]) without ridge (the partial a shard contributes).  F is nrow x k row-major
 * (k x nrow column-major).  use_lists = 0: the rows are hashed inside the kernel; 1: from the mask lists built first. */
SGL_API int sgl_op_mask_gram(sgl_ctx* ctx, const double* F, const double* G, int32_t k, int32_t nrow, int64_t ncols, uint64_t seed,
                uint64_t inv_density, int mask_t, int64_t col_offset, int64_t row_offset, int use_lists, double* out);
/* Right-hand sides of predict (src/singlet.cpp:341-343) for the resident
 * shard: which = 0: B = F * A (F k x nrow, B k x ncol); which = 1: B = F * At;
 * which = 2 / 3: the same two products through the LDS-tiled kernel (k <= 128) that the
 * fit uses, instead of the plain CSC kernel. */
SGL_API int sgl_op_rhs(sgl_ctx* ctx, int which, const double* F, int32_t k, double* B);
/* nnls (src/singlet.cpp:229-250) on ncols independent columns sharing G:
 * B k x ncols (destroyed on the device, not written back), X k x ncols in/out. */
SGL_API int sgl_op_nnls(sgl_ctx* ctx, const double* G, const double* B, double* X, int32_t k, int64_t ncols,
                double L1, double L2, int32_t* sweeps_out);
/* scale (src/singlet.cpp:219-225) and cor (:184-197). */
SGL_API int sgl_op_scale(sgl_ctx* ctx, double* F, int32_t k, int64_t cols, double* d);
SGL_API int sgl_op_cor(sgl_ctx* ctx, const double* x, const double* y, int64_t n, double* out);
/* mse_test (src/singlet.cpp:536-568) on the resident shard with the current factors. */
SGL_API int sgl_op_mse_test(sgl_ctx* ctx, uint64_t seed, uint64_t inv_density, double* out);

/* ------------------------------------------------------------------------
 * 4. Timing (hipEvent based, on the context's stream).
 * ---------------------------------------------------------------------- */
#define SGL_PH_GRAM 0      /* AAt kernels */
#define SGL_PH_RHS_H 1     /* sparse accumulate, H-update (over A) */
#define SGL_PH_NNLS_H 2
#define SGL_PH_RHS_W 3     /* sparse accumulate, W-update (over At) */
#define SGL_PH_NNLS_W 4
#define SGL_PH_SCALE 5     /* row sums, scale, cor, copies */
#define SGL_PH_COMM 6      /* all-reduce callback */
#define SGL_PH_MASK 7      /* masked path: the per-column Gram downdates of predict_mask (src/singlet.cpp:458-463), mask lists */
#define SGL_PH_MSE 8       /* masked path: mse_test (src/singlet.cpp:536-568), one call per trace row */
#define SGL_PH_COUNT 9
/* Enable/disable per-phase timing (costs two event records per kernel group). */
SGL_API int sgl_timing_enable(sgl_ctx* ctx, int on);
/* ms[SGL_PH_COUNT] accumulated since the last reset, calls[SGL_PH_COUNT] launches. */
SGL_API int sgl_timing_get(sgl_ctx* ctx, double* ms, int64_t* calls, int reset);
/* NNLS sweep totals since the last reset: [0] sweeps summed over the columns of
 * the H solves, [1] same for the W solves; [2], [3] sweeps each 64-column wave
 * actually executed (the maximum over its columns), summed over the waves of
 * the H / W solves -- the number that drives the kernel's run time. */
SGL_API int sgl_sweeps_get(sgl_ctx* ctx, int64_t* out4, int reset);
/* HBM layout of the current fit's entry streams (DESIGN.md "Data layout"), for
 * the roofline report: per orientation (A then At) five numbers --
 * entries stored (non-zeros + padding), row tiles T, rows per tile TR, tile
 * ranges R (blockIdx.y slabs), column blocks of 64.  out10 all zero when the
 * fit runs on the plain CSC kernel (k > 128). */
SGL_API int sgl_layout_get(sgl_ctx* ctx, int64_t* out10);
/* How many times derived data of the resident matrix has been (re)written on this context, out4 = entry stream of A, of
 * At, mask lists of the cell side, of the gene side.  A fit at an unchanged rank reuses the streams (sgl_fit_init above);
 * a masked fit (src/singlet.cpp:436-466, the test set rng(seed).draw(inv_density, ...)) whose (seed, inv_density) was
 * among the last SGL_MASK_KEEP + 1 (default 3: R's n_replicates, R/ard_nmf.R:20) reuses the lists of drawn entries --
 * the mask does not depend on the rank. */
SGL_API int sgl_layout_builds(sgl_ctx* ctx, int64_t* out4);
/* Drawn (cell, gene) pairs of the mask the current fit runs under -- the entries predict_mask leaves out and mse_test scores
 * (src/singlet.cpp:445-448: rng(seed).draw(inv_density, ...) true), as this shard's lists hold them: out2 = pairs listed
 * per cell (H-update), per gene (W-update); 0 where the lists are not built (no masked pass yet, or the hashing kernels run).
 * One masked iteration forms out2[0] + out2[1] rank-one downdates w_r w_r^T: the work unit of the measurement in bench.py. */
SGL_API int sgl_mask_pairs(sgl_ctx* ctx, int64_t* out2);
/* Host wall-clock seconds of the calling thread's LAST one-shot call (sgl_c_nmf / sgl_c_ard_nmf: what an R caller of
 * run_nmf -> .Call(_singlet_c_nmf), R/run_nmf.R:39-59, src/RcppExports.cpp:98-116, waits for around the iterations).  Up to n of
 * out[0] host -> device copies of the dgCMatrix slots, [1] validation kernels (ascending rows, finite values), [2] device transpose
 * (At = NULL), [3] sgl_fit_init (entry streams, w), [4] the ALS loop, [5] factors back to the host, [6] bytes copied in,
 * [7] 1.0 when SINGLET_HIP_CACHE served the resident matrix of the previous call (nothing uploaded), [8] the whole call. */
SGL_API int sgl_call_times_get(double* out, int32_t n);
/* Device memory the library keeps for reuse (round 6): blocks of 64 MB and more that a context frees -- the matrix slots, the
 * entry streams, the factors of a one-shot call -- are cached per device and serve the next request they fit instead of going
 * back to the driver, whose hipMalloc right after such a free takes seconds (3 s for config 3's 22 GB streams: as long as the
 * fit; R's ard_nmf makes tens of such calls, R/ard_nmf.R:95-160).  *cached_bytes = bytes cached on the current device.  They
 * stay reserved by the process until sgl_cache_release(); SGL_POOL=0 in the environment switches the caching off,
 * SGL_POOL_MAX_GB caps it (default: 70 % of the device memory). */
SGL_API int sgl_pool_info(int64_t* cached_bytes);

#ifdef __cplusplus
}
#endif
#endif /* SINGLET_HIP_H */
