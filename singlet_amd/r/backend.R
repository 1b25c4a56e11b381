# backend.R -- opt-in switch for the HIP back end (source()d or added to the package's R/).
#
# With options(singlet.backend = "hip") (or SINGLET_BACKEND=hip in the environment) the
# wrappers of R/RcppExports.R:4-6, 20-38, 78-88 are rebound to the shim's .Call symbols; with the option
# unset nothing changes and the package's own OpenMP code runs.  run_nmf / ard_nmf /
# cross_validate_nmf / RunNMF / project_model call these wrappers by name, so they need no edit.

singlet_hip_enable <- function(shim = Sys.getenv("SINGLET_HIP_SHIM", "singlet_hip_shim.so")) {
  dll <- dyn.load(shim)
  ns <- asNamespace("singlet")
  rebind <- function(name, fn) {
    unlockBinding(name, ns)
    assign(name, fn, envir = ns)
    lockBinding(name, ns)
  }
  rebind("c_nmf", function(A, At, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w)
    .Call(dll[["_singlet_c_nmf"]], A, At, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w))
  rebind("c_ard_nmf", function(A, At, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density, overfit_threshold, trace_test_mse)
    .Call(dll[["_singlet_c_ard_nmf"]], A, At, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density,
          overfit_threshold, trace_test_mse))
  rebind("c_linked_nmf", function(A, At, tol, maxit, verbose, L1, L2, threads, w, link_h, link_w)
    .Call(dll[["_singlet_c_linked_nmf"]], A, At, tol, maxit, verbose, L1, L2, threads, w, link_h, link_w))
  rebind("c_nmf_dense", function(A, At, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w)
    .Call(dll[["_singlet_c_nmf_dense"]], A, At, tol, maxit, verbose, L1_w, L1_h, L2_w, L2_h, threads, w))
  rebind("c_nmf_sparse_list", function(A_, At_, tol, maxit, verbose, L1, L2, threads, w)
    .Call(dll[["_singlet_c_nmf_sparse_list"]], A_, At_, tol, maxit, verbose, L1, L2, threads, w))
  rebind("c_ard_nmf_sparse_list", function(A_, At_, tol, maxit, verbose, L1, L2, threads, w, rng_seed, inv_density,
                                            overfit_threshold, trace_test_mse)
    .Call(dll[["_singlet_c_ard_nmf_sparse_list"]], A_, At_, tol, maxit, verbose, L1, L2, threads, w, rng_seed, inv_density,
          overfit_threshold, trace_test_mse))
  rebind("c_ard_nmf_dense", function(A, At, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density, overfit_threshold,
                                      trace_test_mse)
    .Call(dll[["_singlet_c_ard_nmf_dense"]], A, At, tol, maxit, verbose, L1, L2, threads, w, seed, inv_density,
          overfit_threshold, trace_test_mse))
  rebind("c_project_model", function(A, w, L1, L2, threads)
    .Call(dll[["_singlet_c_project_model"]], A, w, L1, L2, threads))
  rebind("Rcpp_predict", function(A, w, L1, L2, threads)
    .Call(dll[["_singlet_Rcpp_predict"]], A, w, L1, L2, threads))
  # R/RunNMF.R:86-93 re-weights the matrix by group before the fit (R/RcppExports.R: weight_by_split(A_, split_by, n_groups))
  rebind("weight_by_split", function(A_, split_by, n_groups)
    .Call(dll[["_singlet_weight_by_split"]], A_, split_by, n_groups))
  invisible(TRUE)
}

.singlet_hip_onload <- function() {
  want <- getOption("singlet.backend", Sys.getenv("SINGLET_BACKEND", ""))
  if (identical(want, "hip")) singlet_hip_enable()
}
