"""ctypes binding of libsinglet_hip.so (the C ABI of include/singlet_hip.h).

The product path: there is no CPU fallback.  If the shared library is missing
or no gfx950 device is usable, calls raise (SingletHipError) instead of
silently computing somewhere else.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGL_LIB_PATH", os.path.join(_HERE, "libsinglet_hip.so"))  # override: kernel ablation builds only

SGL_PH_NAMES = ("gram", "rhs_h", "nnls_h", "rhs_w", "nnls_w", "scale", "comm", "mask", "mse")
SGL_PH_COUNT = len(SGL_PH_NAMES)

f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)

LOG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_double, C.c_double)
POLL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)


class Callbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("log", LOG_FN), ("poll", POLL_FN)]


class SingletHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsinglet_hip error %d: %s" % (code, msg))
        self.code = code


# every symbol include/singlet_hip.h declares: name -> (restype, argtypes)
_CSC = [f64p, i32p, i32p]
_PPF, _PPI = C.POINTER(f64p), C.POINTER(i32p)
_LIST = [C.c_int32, _PPF, _PPI, _PPI, i32p]
_CB = C.POINTER(Callbacks)
SIGNATURES = {
    "sgl_last_error": (C.c_char_p, []),
    "sgl_abi_version": (C.c_int, []),
    "sgl_device_count": (C.c_int, []),
    "sgl_cache_release": (C.c_int, []),
    "sgl_c_nmf": (C.c_int, _CSC + _CSC + [C.c_int32, C.c_int32, C.c_double, C.c_uint16, C.c_int, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.c_uint16, f64p, C.c_int32, f64p, f64p,
                                          f64p, i32p, f64p, _CB]),
    "sgl_c_ard_nmf": (C.c_int, _CSC + _CSC + [C.c_int32, C.c_int32, C.c_double, C.c_uint16, C.c_int, C.c_double,
                                              C.c_double, C.c_uint16, f64p, C.c_int32, C.c_uint64, C.c_uint64,
                                              C.c_double, C.c_uint16, f64p, f64p, f64p, f64p, i32p, f64p, f64p, i32p,
                                              _CB]),
    "sgl_c_nmf_dense": (C.c_int, [f64p, C.c_int32, C.c_int32, C.c_double, C.c_uint16, C.c_int, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_uint16, f64p, C.c_int32, f64p, f64p, f64p, i32p, f64p, _CB]),
    "sgl_c_ard_nmf_dense": (C.c_int, [f64p, C.c_int32, C.c_int32, C.c_double, C.c_uint16, C.c_int, C.c_double, C.c_double,
                                      C.c_uint16, f64p, C.c_int32, C.c_uint64, C.c_uint64, C.c_double, C.c_uint16, f64p, f64p,
                                      f64p, f64p, i32p, f64p, f64p, i32p, _CB]),
    "sgl_c_nmf_sparse_list": (C.c_int, _LIST + _LIST + [C.c_int32, C.c_double, C.c_uint16, C.c_int, C.c_double, C.c_double,
                                                        C.c_uint16, f64p, C.c_int32, f64p, f64p, f64p, i32p, f64p, _CB]),
    "sgl_c_ard_nmf_sparse_list": (C.c_int, _LIST + _LIST + [C.c_int32, C.c_double, C.c_uint16, C.c_int, C.c_double, C.c_double,
                                                            C.c_uint16, f64p, C.c_int32, C.c_uint64, C.c_uint64, C.c_double,
                                                            C.c_uint16, f64p, f64p, f64p, f64p, i32p, f64p, f64p, i32p, _CB]),
    "sgl_upload_csc_list": (C.c_int, [C.c_void_p] + _LIST + _LIST + [C.c_int32, C.c_int64, C.c_int64]),
    "sgl_c_linked_nmf": (C.c_int, _CSC + _CSC + [C.c_int32, C.c_int32, C.c_double, C.c_uint16, C.c_int, C.c_double,
                                                 C.c_double, C.c_uint16, f64p, C.c_int32, f64p, C.c_int32, C.c_int32, f64p,
                                                 C.c_int32, C.c_int32, f64p, f64p, f64p, i32p, f64p, _CB]),
    "sgl_c_project_model": (C.c_int, _CSC + [C.c_int32, C.c_int32, f64p, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                             C.c_uint16, f64p, f64p]),
    "sgl_rcpp_predict": (C.c_int, _CSC + [C.c_int32, C.c_int32, f64p, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                          C.c_uint16, f64p]),
    "sgl_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "sgl_destroy": (C.c_int, [C.c_void_p]),
    "sgl_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sgl_upload_csc": (C.c_int, [C.c_void_p] + _CSC + _CSC + [C.c_int32, C.c_int32, C.c_int64, C.c_int64]),
    "sgl_upload_dense": (C.c_int, [C.c_void_p, f64p, C.c_int32, C.c_int32]),
    "sgl_synth_csc": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, f64p, C.c_int32, C.c_int64, C.c_int32, C.c_int64]),
    "sgl_synth_csc_skewed": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, f64p, C.c_int32, C.c_int64, C.c_int32, C.c_int64,
                                       f64p, f64p]),
    "sgl_dims": (C.c_int, [C.c_void_p, i32p, i32p, i64p]),
    "sgl_download_csc": (C.c_int, [C.c_void_p, C.c_int, f64p, i32p, i64p]),
    "sgl_log_normalize": (C.c_int, [C.c_void_p, C.c_double]),
    "sgl_weight_by_split": (C.c_int, [C.c_void_p, i32p, C.c_int32]),
    "sgl_c_weight_by_split": (C.c_int, _CSC + [C.c_int32, C.c_int32, i32p, C.c_int32, f64p]),
    "sgl_fit_init": (C.c_int, [C.c_void_p, C.c_int32, f64p, C.c_uint64]),
    "sgl_set_links": (C.c_int, [C.c_void_p, f64p, C.c_int32, C.c_int32, f64p, C.c_int32, C.c_int32]),
    "sgl_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "sgl_step_begin": (C.c_int, [C.c_void_p]),
    "sgl_step_h": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "sgl_step_scale_h": (C.c_int, [C.c_void_p]),
    "sgl_step_w": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "sgl_step_h_masked": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_uint64, C.c_uint64]),
    "sgl_step_w_masked": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_uint64, C.c_uint64]),
    "sgl_step_scale_w": (C.c_int, [C.c_void_p, f64p]),
    "sgl_nmf_run": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, i32p,
                              f64p, _CB]),
    "sgl_ard_run": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                              C.c_double, C.c_int32, f64p, i32p, f64p, f64p, i32p, i32p, _CB]),
    "sgl_project_run": (C.c_int, [C.c_void_p, C.c_double, C.c_double]),
    "sgl_nmf_iterate": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, f64p]),
    "sgl_multi_create": (C.c_int, [C.c_int, i32p, C.POINTER(C.c_void_p)]),
    "sgl_multi_destroy": (C.c_int, [C.c_void_p]),
    "sgl_multi_size": (C.c_int, [C.c_void_p]),
    "sgl_multi_ctx": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "sgl_multi_upload_csc": (C.c_int, [C.c_void_p] + _CSC + [C.c_int32, C.c_int32]),
    "sgl_multi_synth_csc": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, f64p, C.c_int32, C.c_int64]),
    "sgl_multi_fit_init": (C.c_int, [C.c_void_p, C.c_int32, f64p, C.c_uint64]),
    "sgl_multi_set_links": (C.c_int, [C.c_void_p, f64p, C.c_int32, C.c_int32, f64p, C.c_int32, C.c_int32]),
    "sgl_multi_iterate": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, f64p]),
    "sgl_multi_nmf_run": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                    i32p, f64p, _CB]),
    "sgl_multi_ard_run": (C.c_int, [C.c_void_p, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                                    C.c_double, C.c_int32, f64p, i32p, f64p, f64p, i32p, i32p, _CB]),
    "sgl_multi_get_factors": (C.c_int, [C.c_void_p, f64p, f64p, f64p]),
    "sgl_comm_unique_id": (C.c_int, [C.c_void_p]),
    "sgl_comm_init_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "sgl_comm_available": (C.c_int, [C.c_char_p, C.c_int]),
    "sgl_comm_info": (C.c_int, [C.c_void_p, i32p, i32p, C.c_char_p, C.c_int]),
    "sgl_split_cells_by_nnz": (C.c_int, [i32p, C.c_int32, C.c_int, i64p]),
    "sgl_get_factors": (C.c_int, [C.c_void_p, f64p, f64p, f64p]),
    "sgl_set_factors": (C.c_int, [C.c_void_p, f64p, f64p, f64p]),
    "sgl_op_rand": (C.c_int, [C.c_void_p, C.c_uint64, u64p, u64p, C.c_int64, u64p]),
    "sgl_op_mask": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, u8p]),
    "sgl_op_gram": (C.c_int, [C.c_void_p, f64p, C.c_int32, C.c_int64, f64p]),
    "sgl_op_rhs": (C.c_int, [C.c_void_p, C.c_int, f64p, C.c_int32, f64p]),
    "sgl_op_nnls": (C.c_int, [C.c_void_p, f64p, f64p, f64p, C.c_int32, C.c_int64, C.c_double, C.c_double, i32p]),
    "sgl_op_mask_gram": (C.c_int, [C.c_void_p, f64p, f64p, C.c_int32, C.c_int32, C.c_int64, C.c_uint64, C.c_uint64, C.c_int, C.c_int64,
                                   C.c_int64, C.c_int, f64p]),
    "sgl_op_scale": (C.c_int, [C.c_void_p, f64p, C.c_int32, C.c_int64, f64p]),
    "sgl_op_cor": (C.c_int, [C.c_void_p, f64p, f64p, C.c_int64, f64p]),
    "sgl_op_mse_test": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, f64p]),
    "sgl_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "sgl_timing_get": (C.c_int, [C.c_void_p, f64p, i64p, C.c_int]),
    "sgl_sweeps_get": (C.c_int, [C.c_void_p, i64p, C.c_int]),
    "sgl_layout_get": (C.c_int, [C.c_void_p, i64p]),
    "sgl_layout_builds": (C.c_int, [C.c_void_p, i64p]),
    "sgl_mask_pairs": (C.c_int, [C.c_void_p, i64p]),
    "sgl_call_times_get": (C.c_int, [f64p, C.c_int32]),
    "sgl_pool_info": (C.c_int, [i64p]),
}

_lib = None


def load():
    """Load libsinglet_hip.so and bind every declared symbol; raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SingletHipError(-2, "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                      "or `make -C singlet_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        override = "SGL_LIB_PATH" in os.environ   # an A/B build of an older tree may lack the newest symbols: bind what it has
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)  # AttributeError if the library does not export it
            except AttributeError:
                if override:
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        msg = load().sgl_last_error()
        raise SingletHipError(rc, msg.decode("utf-8", "replace") if msg else "")
    return rc


def ptr(a, t):
    return None if a is None else a.ctypes.data_as(t)
