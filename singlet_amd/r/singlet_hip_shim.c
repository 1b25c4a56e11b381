/*
 * singlet_hip_shim.c -- the only R-aware C file of the HIP back end.
 *
 * It provides .Call entry points with the SAME names and arity as the Rcpp glue
 * of the reference (src/RcppExports.cpp:98-116 _singlet_c_nmf,
 * :284-304 _singlet_c_ard_nmf, _singlet_c_project_model, _singlet_Rcpp_predict;
 * registration table :449-477), so that R/RcppExports.R:20-30, 78-80 and
 * everything above it (run_nmf, ard_nmf, cross_validate_nmf, RunNMF,
 * project_model) work unchanged.  Each entry point pulls raw pointers out of
 * the dgCMatrix slots (what inst/include/singlet.h:108-127 does through Rcpp),
 * allocates the result objects, and forwards to libsinglet_hip.so
 * (include/singlet_hip.h).  It uses the plain R C API only: no Rcpp, no Eigen.
 *
 * NOT COMPILED IN THIS REPOSITORY'S CI: the build image has neither R nor its
 * headers.  Build where R exists with
 *     R CMD SHLIB singlet_hip_shim.c -I<repo>/include -L<repo>/singlet_amd -lsinglet_hip
 * (or add the file to the package's src/ and the two flags to src/Makevars;
 * see INTEGRATION.md).  All numerical testing of the library goes through the
 * same C ABI from Python (tests/).
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <R_ext/Utils.h>
#include <math.h>
#include <stdint.h>

#include "singlet_hip.h"

/* ---- argument helpers -------------------------------------------------- */
typedef struct {
    const double* x;
    const int* i;
    const int* p;
    int nrow, ncol;
} dgc_view;

/* Zero-copy view of a Matrix::dgCMatrix; errors like the reference's Exporter
 * (inst/include/singlet.h:116-117) when a slot is missing. */
static dgc_view view_dgc(SEXP s, const char* what) {
    dgc_view v;
    SEXP names[4] = {Rf_install("x"), Rf_install("i"), Rf_install("p"), Rf_install("Dim")};
    for (int q = 0; q < 4; ++q)
        if (!R_has_slot(s, names[q])) Rf_error("%s: not a dgCMatrix (missing slot)", what);
    SEXP x = R_do_slot(s, names[0]), i = R_do_slot(s, names[1]), p = R_do_slot(s, names[2]), dim = R_do_slot(s, names[3]);
    if (TYPEOF(x) != REALSXP || TYPEOF(i) != INTSXP || TYPEOF(p) != INTSXP || TYPEOF(dim) != INTSXP || XLENGTH(dim) != 2)
        Rf_error("%s: not a dgCMatrix (slot types)", what);
    v.x = REAL(x);
    v.i = INTEGER(i);
    v.p = INTEGER(p);
    v.nrow = INTEGER(dim)[0];
    v.ncol = INTEGER(dim)[1];
    return v;
}

static void fail_if(int rc) {
    if (rc != SGL_OK) Rf_error("singlet HIP back end: %s", sgl_last_error());
}

/* Interrupts: R_CheckUserInterrupt() long-jumps, which must not unwind through
 * the library while kernels are in flight; test for a pending interrupt inside
 * R_ToplevelExec and report it as a return value instead.  The library stops
 * at the next polling point, frees its device memory and returns SGL_EINTR. */
static void check_interrupt_fn(void* dummy) { (void)dummy; R_CheckUserInterrupt(); }
static int poll_cb(void* user) { (void)user; return R_ToplevelExec(check_interrupt_fn, NULL) == FALSE; }

/* Trace lines exactly as the reference prints them (src/singlet.cpp:661-662, 1115-1127). */
static void log_nmf(void* user, int iter, double tol, double overfit) {
    (void)user; (void)overfit;
    Rprintf("%4d | %8.2e\n", iter, tol);
}
static void log_ard(void* user, int iter, double tol, double overfit) {
    (void)user;
    if (ISNAN(overfit)) Rprintf("%4d | %8.2e | %8s\n", iter, tol, "-");
    else Rprintf("%4d | %8.2e | %8.2e\n", iter, tol, overfit);
}

static SEXP named_list(int n, const char** names, SEXP* values) {
    SEXP out = PROTECT(Rf_allocVector(VECSXP, n)), nm = PROTECT(Rf_allocVector(STRSXP, n));
    for (int q = 0; q < n; ++q) {
        SET_VECTOR_ELT(out, q, values[q]);
        SET_STRING_ELT(nm, q, Rf_mkChar(names[q]));
    }
    Rf_setAttrib(out, R_NamesSymbol, nm);
    UNPROTECT(2);
    return out;
}

/* ---- c_nmf(A, At, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w) ---- */
SEXP _singlet_c_nmf(SEXP A_, SEXP At_, SEXP tol_, SEXP maxit_, SEXP verbose_, SEXP L1w_, SEXP L1h_, SEXP L2w_, SEXP L2h_,
                    SEXP threads_, SEXP w_) {
    dgc_view A = view_dgc(A_, "A"), At = view_dgc(At_, "At");
    if (!Rf_isMatrix(w_) || TYPEOF(w_) != REALSXP) Rf_error("w must be a numeric matrix");
    const int k = Rf_nrows(w_), m = Rf_ncols(w_);
    if (m != A.nrow) Rf_error("w must be k x nrow(A)");
    const int verbose = Rf_asLogical(verbose_);
    const int maxit = Rf_asInteger(maxit_);
    SEXP w = PROTECT(Rf_allocMatrix(REALSXP, k, A.nrow)), d = PROTECT(Rf_allocVector(REALSXP, k)),
         h = PROTECT(Rf_allocMatrix(REALSXP, k, A.ncol));
    sgl_callbacks cb = {NULL, verbose ? log_nmf : NULL, poll_cb};
    if (verbose) Rprintf("\n%4s | %8s \n---------------\n", "iter", "tol");
    int n_iter = 0;
    int rc = sgl_c_nmf(A.x, A.i, A.p, At.x, At.i, At.p, A.nrow, A.ncol, Rf_asReal(tol_), (uint16_t)maxit, verbose,
                       Rf_asReal(L1w_), Rf_asReal(L1h_), Rf_asReal(L2w_), Rf_asReal(L2h_), (uint16_t)Rf_asInteger(threads_),
                       REAL(w_), k, REAL(w), REAL(d), REAL(h), &n_iter, NULL, &cb);
    if (rc == SGL_EINTR) { UNPROTECT(3); Rf_onintr(); }
    fail_if(rc);
    const char* names[3] = {"w", "d", "h"};
    SEXP vals[3] = {w, d, h};
    SEXP out = named_list(3, names, vals);
    UNPROTECT(3);
    return out;
}

/* ---- c_ard_nmf(A, At, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density,
 *                overfit_threshold, trace_test_mse) -------------------------------- */
SEXP _singlet_c_ard_nmf(SEXP A_, SEXP At_, SEXP tol_, SEXP maxit_, SEXP verbose_, SEXP L1_, SEXP L2_, SEXP threads_, SEXP w_,
                        SEXP seed_, SEXP invd_, SEXP thr_, SEXP trace_) {
    dgc_view A = view_dgc(A_, "A"), At = view_dgc(At_, "At");
    if (!Rf_isMatrix(w_) || TYPEOF(w_) != REALSXP) Rf_error("w must be a numeric matrix");
    const int k = Rf_nrows(w_);
    if (Rf_ncols(w_) != A.nrow) Rf_error("w must be k x nrow(A)");
    const int verbose = Rf_asLogical(verbose_);
    const int maxit = Rf_asInteger(maxit_);
    SEXP w = PROTECT(Rf_allocMatrix(REALSXP, k, A.nrow)), d = PROTECT(Rf_allocVector(REALSXP, k)),
         h = PROTECT(Rf_allocMatrix(REALSXP, k, A.ncol));
    double* t_mse = (double*)R_alloc(maxit + 1, sizeof(double));
    double* t_tol = (double*)R_alloc(maxit + 1, sizeof(double));
    double* t_sco = (double*)R_alloc(maxit + 1, sizeof(double));
    int* t_it = (int*)R_alloc(maxit + 1, sizeof(int));
    int n_trace = 0;
    sgl_callbacks cb = {NULL, verbose ? log_ard : NULL, poll_cb};
    if (verbose) Rprintf("\n%4s | %8s | %8s \n---------------------------\n", "iter", "tol", "overfit");
    int rc = sgl_c_ard_nmf(A.x, A.i, A.p, At.x, At.i, At.p, A.nrow, A.ncol, Rf_asReal(tol_), (uint16_t)maxit, verbose,
                           Rf_asReal(L1_), Rf_asReal(L2_), (uint16_t)Rf_asInteger(threads_), REAL(w_), k,
                           (uint64_t)Rf_asReal(seed_), (uint64_t)Rf_asReal(invd_), Rf_asReal(thr_),
                           (uint16_t)Rf_asInteger(trace_), REAL(w), REAL(d), REAL(h), t_mse, t_it, t_tol, t_sco, &n_trace, &cb);
    if (rc == SGL_EINTR) { UNPROTECT(3); Rf_onintr(); }
    fail_if(rc);
    SEXP v_mse = PROTECT(Rf_allocVector(REALSXP, n_trace)), v_it = PROTECT(Rf_allocVector(INTSXP, n_trace)),
         v_tol = PROTECT(Rf_allocVector(REALSXP, n_trace)), v_sco = PROTECT(Rf_allocVector(REALSXP, n_trace));
    for (int q = 0; q < n_trace; ++q) {
        REAL(v_mse)[q] = t_mse[q];
        INTEGER(v_it)[q] = t_it[q];
        REAL(v_tol)[q] = t_tol[q];
        REAL(v_sco)[q] = t_sco[q];
    }
    /* element names and order of src/singlet.cpp:1144-1151 */
    const char* names[7] = {"w", "d", "h", "test_mse", "iter", "tol", "score_overfit"};
    SEXP vals[7] = {w, d, h, v_mse, v_it, v_tol, v_sco};
    SEXP out = named_list(7, names, vals);
    UNPROTECT(7);
    return out;
}

/* ---- c_linked_nmf(A, At, tol, maxit, verbose, L1, L2, threads, w, link_h, link_w) ---- *
 * (src/singlet.cpp:1059-1086; R/RunLNMF.R:60).  A link whose column count does not match its
 * side is ignored by the library, as by the reference. */
SEXP _singlet_c_linked_nmf(SEXP A_, SEXP At_, SEXP tol_, SEXP maxit_, SEXP verbose_, SEXP L1_, SEXP L2_, SEXP threads_, SEXP w_,
                           SEXP link_h_, SEXP link_w_) {
    dgc_view A = view_dgc(A_, "A"), At = view_dgc(At_, "At");
    if (!Rf_isMatrix(w_) || TYPEOF(w_) != REALSXP) Rf_error("w must be a numeric matrix");
    if (!Rf_isMatrix(link_h_) || TYPEOF(link_h_) != REALSXP || !Rf_isMatrix(link_w_) || TYPEOF(link_w_) != REALSXP)
        Rf_error("link_h and link_w must be numeric matrices");
    const int k = Rf_nrows(w_);
    if (Rf_ncols(w_) != A.nrow) Rf_error("w must be k x nrow(A)");
    const int verbose = Rf_asLogical(verbose_);
    SEXP w = PROTECT(Rf_allocMatrix(REALSXP, k, A.nrow)), d = PROTECT(Rf_allocVector(REALSXP, k)),
         h = PROTECT(Rf_allocMatrix(REALSXP, k, A.ncol));
    sgl_callbacks cb = {NULL, verbose ? log_nmf : NULL, poll_cb};
    if (verbose) Rprintf("\n%4s | %8s \n---------------\n", "iter", "tol");
    int n_iter = 0;
    int rc = sgl_c_linked_nmf(A.x, A.i, A.p, At.x, At.i, At.p, A.nrow, A.ncol, Rf_asReal(tol_), (uint16_t)Rf_asInteger(maxit_),
                              verbose, Rf_asReal(L1_), Rf_asReal(L2_), (uint16_t)Rf_asInteger(threads_), REAL(w_), k,
                              REAL(link_h_), Rf_nrows(link_h_), Rf_ncols(link_h_), REAL(link_w_), Rf_nrows(link_w_),
                              Rf_ncols(link_w_), REAL(w), REAL(d), REAL(h), &n_iter, NULL, &cb);
    if (rc == SGL_EINTR) { UNPROTECT(3); Rf_onintr(); }
    fail_if(rc);
    const char* names[3] = {"w", "d", "h"};
    SEXP vals[3] = {w, d, h};
    SEXP out = named_list(3, names, vals);
    UNPROTECT(3);
    return out;
}

/* ---- c_nmf_dense(A, At, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w) ---- *
 * (src/singlet.cpp:1052-1054; R/run_nmf.R:57).  A is a base numeric matrix; At is not needed. */
SEXP _singlet_c_nmf_dense(SEXP A_, SEXP At_, SEXP tol_, SEXP maxit_, SEXP verbose_, SEXP L1w_, SEXP L1h_, SEXP L2w_, SEXP L2h_,
                          SEXP threads_, SEXP w_) {
    (void)At_;
    if (!Rf_isMatrix(A_) || TYPEOF(A_) != REALSXP) Rf_error("A must be a numeric matrix");
    if (!Rf_isMatrix(w_) || TYPEOF(w_) != REALSXP) Rf_error("w must be a numeric matrix");
    const int m = Rf_nrows(A_), n = Rf_ncols(A_), k = Rf_nrows(w_);
    if (Rf_ncols(w_) != m) Rf_error("w must be k x nrow(A)");
    const int verbose = Rf_asLogical(verbose_);
    SEXP w = PROTECT(Rf_allocMatrix(REALSXP, k, m)), d = PROTECT(Rf_allocVector(REALSXP, k)),
         h = PROTECT(Rf_allocMatrix(REALSXP, k, n));
    sgl_callbacks cb = {NULL, verbose ? log_nmf : NULL, poll_cb};
    if (verbose) Rprintf("\n%4s | %8s \n---------------\n", "iter", "tol");
    int n_iter = 0;
    int rc = sgl_c_nmf_dense(REAL(A_), m, n, Rf_asReal(tol_), (uint16_t)Rf_asInteger(maxit_), verbose, Rf_asReal(L1w_),
                             Rf_asReal(L1h_), Rf_asReal(L2w_), Rf_asReal(L2h_), (uint16_t)Rf_asInteger(threads_), REAL(w_), k,
                             REAL(w), REAL(d), REAL(h), &n_iter, NULL, &cb);
    if (rc == SGL_EINTR) { UNPROTECT(3); Rf_onintr(); }
    fail_if(rc);
    const char* names[3] = {"w", "d", "h"};
    SEXP vals[3] = {w, d, h};
    SEXP out = named_list(3, names, vals);
    UNPROTECT(3);
    return out;
}

/* ---- c_project_model(A, w, L1, L2, threads) -> list(h, d) ---------------- */
SEXP _singlet_c_project_model(SEXP A_, SEXP w_, SEXP L1_, SEXP L2_, SEXP threads_) {
    dgc_view A = view_dgc(A_, "A");
    if (!Rf_isMatrix(w_) || TYPEOF(w_) != REALSXP) Rf_error("w must be a numeric matrix");
    const int wr = Rf_nrows(w_), wc = Rf_ncols(w_);
    const int k = (wr == A.nrow) ? wc : wr;
    SEXP h = PROTECT(Rf_allocMatrix(REALSXP, k, A.ncol)), d = PROTECT(Rf_allocVector(REALSXP, k));
    fail_if(sgl_c_project_model(A.x, A.i, A.p, A.nrow, A.ncol, REAL(w_), wr, wc, Rf_asReal(L1_), Rf_asReal(L2_),
                                (uint16_t)Rf_asInteger(threads_), REAL(h), REAL(d)));
    const char* names[2] = {"h", "d"};
    SEXP vals[2] = {h, d};
    SEXP out = named_list(2, names, vals);
    UNPROTECT(2);
    return out;
}

/* ---- Rcpp_predict(A, w, L1, L2, threads) -> h ---------------------------- */
SEXP _singlet_Rcpp_predict(SEXP A_, SEXP w_, SEXP L1_, SEXP L2_, SEXP threads_) {
    dgc_view A = view_dgc(A_, "A");
    if (!Rf_isMatrix(w_) || TYPEOF(w_) != REALSXP) Rf_error("w must be a numeric matrix");
    const int wr = Rf_nrows(w_), wc = Rf_ncols(w_);
    const int k = (wr == A.nrow && wc != A.nrow) ? wc : wr;
    SEXP h = PROTECT(Rf_allocMatrix(REALSXP, k, A.ncol));
    fail_if(sgl_rcpp_predict(A.x, A.i, A.p, A.nrow, A.ncol, REAL(w_), wr, wc, Rf_asReal(L1_), Rf_asReal(L2_),
                             (uint16_t)Rf_asInteger(threads_), REAL(h)));
    UNPROTECT(1);
    return h;
}

/* ---- registration (stand-alone build: library(singletHip) style) ---------- *
 * When the file is compiled INTO the singlet package instead, drop this table
 * and keep the reference's own CallEntries (src/RcppExports.cpp:449-477): the
 * four symbols above then simply replace the four Rcpp-generated ones. */
/* ---- chunk lists: R lists of dgCMatrix column chunks (R/ard_nmf.R:114, 181) ------------------------- */
typedef struct {
    int n;
    const double** x;
    const int** i;
    const int** p;
    int* ncol;
    int nrow;
} dgc_list;

static dgc_list view_dgc_list(SEXP lst, const char* what) {
    dgc_list L;
    if (TYPEOF(lst) != VECSXP || XLENGTH(lst) < 1) Rf_error("%s: not a non-empty list of dgCMatrix", what);
    L.n = (int)XLENGTH(lst);
    L.x = (const double**)R_alloc(L.n, sizeof(double*));
    L.i = (const int**)R_alloc(L.n, sizeof(int*));
    L.p = (const int**)R_alloc(L.n, sizeof(int*));
    L.ncol = (int*)R_alloc(L.n, sizeof(int));
    L.nrow = 0;
    for (int q = 0; q < L.n; ++q) {
        dgc_view v = view_dgc(VECTOR_ELT(lst, q), what);
        if (q == 0) L.nrow = v.nrow;
        else if (v.nrow != L.nrow) Rf_error("%s: chunks differ in their number of rows", what);
        L.x[q] = v.x; L.i[q] = v.i; L.p[q] = v.p; L.ncol[q] = v.ncol;
    }
    return L;
}

/* h is allocated k x nrow(At[[1]]) as the reference does (l.723), and the library writes k x (total columns of the A
 * chunks): the two must agree or the write would run past the R vector */
static void check_list_cells(const dgc_list* A, int n_from_At) {
    long tot = 0;
    for (int q = 0; q < A->n; ++q) tot += A->ncol[q];
    if (tot != (long)n_from_At) Rf_error("the chunks of A hold %ld columns in all but the chunks of At have %d rows", tot, n_from_At);
}

/* ---- c_nmf_sparse_list(A_, At_, tol, maxit, verbose, L1, L2, threads, w)  (src/singlet.cpp:715-743) ---- */
SEXP _singlet_c_nmf_sparse_list(SEXP A_, SEXP At_, SEXP tol_, SEXP maxit_, SEXP verbose_, SEXP L1_, SEXP L2_, SEXP threads_,
                                SEXP w_) {
    dgc_list A = view_dgc_list(A_, "A"), At = view_dgc_list(At_, "At");
    if (!Rf_isMatrix(w_) || TYPEOF(w_) != REALSXP) Rf_error("w must be a numeric matrix");
    const int k = Rf_nrows(w_);
    if (Rf_ncols(w_) != A.nrow) Rf_error("w must be k x nrow(A)");
    const int n = At.nrow;   /* n = At[0].rows(), l.723 */
    check_list_cells(&A, n);
    const int verbose = Rf_asLogical(verbose_);
    const int maxit = Rf_asInteger(maxit_);
    SEXP w = PROTECT(Rf_allocMatrix(REALSXP, k, A.nrow)), d = PROTECT(Rf_allocVector(REALSXP, k)),
         h = PROTECT(Rf_allocMatrix(REALSXP, k, n));
    sgl_callbacks cb = {NULL, verbose ? log_nmf : NULL, poll_cb};
    if (verbose) Rprintf("\n%4s | %8s \n---------------\n", "iter", "tol");
    int n_iter = 0;
    int rc = sgl_c_nmf_sparse_list(A.n, A.x, A.i, A.p, A.ncol, At.n, At.x, At.i, At.p, At.ncol, A.nrow, Rf_asReal(tol_),
                                   (uint16_t)maxit, verbose, Rf_asReal(L1_), Rf_asReal(L2_), (uint16_t)Rf_asInteger(threads_),
                                   REAL(w_), k, REAL(w), REAL(d), REAL(h), &n_iter, NULL, &cb);
    if (rc == SGL_EINTR) { UNPROTECT(3); Rf_onintr(); }
    fail_if(rc);
    const char* names[3] = {"w", "d", "h"};
    SEXP vals[3] = {w, d, h};
    SEXP out = named_list(3, names, vals);
    UNPROTECT(3);
    return out;
}

/* shared tail of the three c_ard_* entry points: trace vectors -> the reference's 7-element list */
static SEXP ard_result(SEXP w, SEXP d, SEXP h, int n_trace, const double* t_mse, const int* t_it, const double* t_tol,
                       const double* t_sco) {
    SEXP v_mse = PROTECT(Rf_allocVector(REALSXP, n_trace)), v_it = PROTECT(Rf_allocVector(INTSXP, n_trace)),
         v_tol = PROTECT(Rf_allocVector(REALSXP, n_trace)), v_sco = PROTECT(Rf_allocVector(REALSXP, n_trace));
    for (int q = 0; q < n_trace; ++q) {
        REAL(v_mse)[q] = t_mse[q];
        INTEGER(v_it)[q] = t_it[q];
        REAL(v_tol)[q] = t_tol[q];
        REAL(v_sco)[q] = t_sco[q];
    }
    const char* names[7] = {"w", "d", "h", "test_mse", "iter", "tol", "score_overfit"};
    SEXP vals[7] = {w, d, h, v_mse, v_it, v_tol, v_sco};
    SEXP out = named_list(7, names, vals);
    UNPROTECT(4);
    return out;
}

/* ---- c_ard_nmf_sparse_list(A_, At_, tol, maxit, verbose, L1, L2, threads, w, rng_seed, inv_density,
 *                            overfit_threshold, trace_test_mse)  (src/singlet.cpp:1162-1234) ---- */
SEXP _singlet_c_ard_nmf_sparse_list(SEXP A_, SEXP At_, SEXP tol_, SEXP maxit_, SEXP verbose_, SEXP L1_, SEXP L2_, SEXP threads_,
                                    SEXP w_, SEXP seed_, SEXP invd_, SEXP thr_, SEXP trace_) {
    dgc_list A = view_dgc_list(A_, "A"), At = view_dgc_list(At_, "At");
    if (!Rf_isMatrix(w_) || TYPEOF(w_) != REALSXP) Rf_error("w must be a numeric matrix");
    const int k = Rf_nrows(w_);
    if (Rf_ncols(w_) != A.nrow) Rf_error("w must be k x nrow(A)");
    const int n = At.nrow;
    check_list_cells(&A, n);
    const int verbose = Rf_asLogical(verbose_);
    const int maxit = Rf_asInteger(maxit_);
    SEXP w = PROTECT(Rf_allocMatrix(REALSXP, k, A.nrow)), d = PROTECT(Rf_allocVector(REALSXP, k)),
         h = PROTECT(Rf_allocMatrix(REALSXP, k, n));
    double* t_mse = (double*)R_alloc(maxit + 1, sizeof(double));
    double* t_tol = (double*)R_alloc(maxit + 1, sizeof(double));
    double* t_sco = (double*)R_alloc(maxit + 1, sizeof(double));
    int* t_it = (int*)R_alloc(maxit + 1, sizeof(int));
    int n_trace = 0;
    sgl_callbacks cb = {NULL, verbose ? log_ard : NULL, poll_cb};
    if (verbose) Rprintf("\n%4s | %8s | %8s \n---------------------------\n", "iter", "tol", "overfit");
    int rc = sgl_c_ard_nmf_sparse_list(A.n, A.x, A.i, A.p, A.ncol, At.n, At.x, At.i, At.p, At.ncol, A.nrow, Rf_asReal(tol_),
                                       (uint16_t)maxit, verbose, Rf_asReal(L1_), Rf_asReal(L2_),
                                       (uint16_t)Rf_asInteger(threads_), REAL(w_), k, (uint64_t)Rf_asReal(seed_),
                                       (uint64_t)Rf_asReal(invd_), Rf_asReal(thr_), (uint16_t)Rf_asInteger(trace_), REAL(w),
                                       REAL(d), REAL(h), t_mse, t_it, t_tol, t_sco, &n_trace, &cb);
    if (rc == SGL_EINTR) { UNPROTECT(3); Rf_onintr(); }
    fail_if(rc);
    SEXP out = ard_result(w, d, h, n_trace, t_mse, t_it, t_tol, t_sco);
    UNPROTECT(3);
    return out;
}

/* ---- c_ard_nmf_dense(A, At, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density, overfit_threshold,
 *                      trace_test_mse)  (src/singlet.cpp:1357-1361); At is not needed ---- */
SEXP _singlet_c_ard_nmf_dense(SEXP A_, SEXP At_, SEXP tol_, SEXP maxit_, SEXP verbose_, SEXP L1_, SEXP L2_, SEXP threads_, SEXP w_,
                              SEXP seed_, SEXP invd_, SEXP thr_, SEXP trace_) {
    (void)At_;
    if (!Rf_isMatrix(A_) || TYPEOF(A_) != REALSXP) Rf_error("A must be a numeric matrix");
    if (!Rf_isMatrix(w_) || TYPEOF(w_) != REALSXP) Rf_error("w must be a numeric matrix");
    const int m = Rf_nrows(A_), n = Rf_ncols(A_), k = Rf_nrows(w_);
    if (Rf_ncols(w_) != m) Rf_error("w must be k x nrow(A)");
    const int verbose = Rf_asLogical(verbose_);
    const int maxit = Rf_asInteger(maxit_);
    SEXP w = PROTECT(Rf_allocMatrix(REALSXP, k, m)), d = PROTECT(Rf_allocVector(REALSXP, k)), h = PROTECT(Rf_allocMatrix(REALSXP, k, n));
    double* t_mse = (double*)R_alloc(maxit + 1, sizeof(double));
    double* t_tol = (double*)R_alloc(maxit + 1, sizeof(double));
    double* t_sco = (double*)R_alloc(maxit + 1, sizeof(double));
    int* t_it = (int*)R_alloc(maxit + 1, sizeof(int));
    int n_trace = 0;
    sgl_callbacks cb = {NULL, verbose ? log_ard : NULL, poll_cb};
    if (verbose) Rprintf("\n%4s | %8s | %8s \n---------------------------\n", "iter", "tol", "overfit");
    int rc = sgl_c_ard_nmf_dense(REAL(A_), m, n, Rf_asReal(tol_), (uint16_t)maxit, verbose, Rf_asReal(L1_), Rf_asReal(L2_),
                                 (uint16_t)Rf_asInteger(threads_), REAL(w_), k, (uint64_t)Rf_asReal(seed_), (uint64_t)Rf_asReal(invd_),
                                 Rf_asReal(thr_), (uint16_t)Rf_asInteger(trace_), REAL(w), REAL(d), REAL(h), t_mse, t_it, t_tol,
                                 t_sco, &n_trace, &cb);
    if (rc == SGL_EINTR) { UNPROTECT(3); Rf_onintr(); }
    fail_if(rc);
    SEXP out = ard_result(w, d, h, n_trace, t_mse, t_it, t_tol, t_sco);
    UNPROTECT(3);
    return out;
}

/* ---- weight_by_split(A_, split_by, n_groups)  (src/singlet.cpp:118-144; R/RunNMF.R:86-93) ----
 * The reference rewrites the x slot of A_ in place and returns the same S4 object (R forces a private copy of A@x
 * before the call, R/RunNMF.R:90-92); so does this. */
SEXP _singlet_weight_by_split(SEXP A_, SEXP split_by_, SEXP n_groups_) {
    dgc_view A = view_dgc(A_, "A");
    if (TYPEOF(split_by_) != INTSXP || XLENGTH(split_by_) != A.ncol) Rf_error("split_by must be an integer vector with one entry per column of A");
    SEXP x = R_do_slot(A_, Rf_install("x"));
    int rc = sgl_c_weight_by_split(A.x, A.i, A.p, A.nrow, A.ncol, INTEGER(split_by_), Rf_asInteger(n_groups_), REAL(x));
    fail_if(rc);
    return A_;
}

static const R_CallMethodDef call_entries[] = {
    {"_singlet_weight_by_split", (DL_FUNC)&_singlet_weight_by_split, 3},
    {"_singlet_c_nmf", (DL_FUNC)&_singlet_c_nmf, 11},
    {"_singlet_c_ard_nmf", (DL_FUNC)&_singlet_c_ard_nmf, 13},
    {"_singlet_c_linked_nmf", (DL_FUNC)&_singlet_c_linked_nmf, 11},
    {"_singlet_c_nmf_dense", (DL_FUNC)&_singlet_c_nmf_dense, 11},
    {"_singlet_c_nmf_sparse_list", (DL_FUNC)&_singlet_c_nmf_sparse_list, 9},
    {"_singlet_c_ard_nmf_sparse_list", (DL_FUNC)&_singlet_c_ard_nmf_sparse_list, 13},
    {"_singlet_c_ard_nmf_dense", (DL_FUNC)&_singlet_c_ard_nmf_dense, 13},
    {"_singlet_c_project_model", (DL_FUNC)&_singlet_c_project_model, 5},
    {"_singlet_Rcpp_predict", (DL_FUNC)&_singlet_Rcpp_predict, 5},
    {NULL, NULL, 0}};

void R_init_singlet_hip_shim(DllInfo* dll) {
    R_registerRoutines(dll, NULL, call_entries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

/* dyn.unload / library detach: the device blocks the library keeps between calls (its pool, and the resident matrix of
 * SINGLET_HIP_CACHE=1) go back to the driver (include/singlet_hip.h: sgl_cache_release). */
void R_unload_singlet_hip_shim(DllInfo* dll) {
    (void)dll;
    (void)sgl_cache_release();
}
