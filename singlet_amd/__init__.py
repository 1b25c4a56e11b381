"""singlet_amd: MI355X (gfx950) engine for singlet's ALS hot path
(c_nmf / c_ard_nmf / c_project_model of src/singlet.cpp) behind a C ABI
(include/singlet_hip.h), plus the Python mirror of the R interface above it."""
from .sparse import dgCMatrix, as_dgCMatrix  # noqa: F401
from .context import Context, Multi, comm_unique_id, comm_available, device_count, split_cells_by_nnz, LEVELS16, SYNTH_SEED  # noqa: F401
from .api import (c_nmf, c_ard_nmf, c_linked_nmf, c_nmf_dense, c_nmf_sparse_list, c_ard_nmf_dense, c_ard_nmf_sparse_list, c_project_model, Rcpp_predict, run_nmf, ard_nmf, cross_validate_nmf,  # noqa: F401
                  GetBestRank, project_model, CVData, PreprocessData, weight_by_split, call_times)
from ._lib import SingletHipError, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
