"""ctypes binding of liborigin_hip.so (include/origin_hip.h).

The library is the product: if it cannot be loaded, or no GPU is usable, every entry point
of this package raises -- there is no CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIBNAME = "liborigin_hip.so"
# (ORIGIN_HIP_LIB: another build of the same library -- kernel variants compiled with
# ORIGIN_HIPCC_FLAGS for A/B timing runs, tools/)
LIBPATH = os.environ.get("ORIGIN_HIP_LIB") or os.path.join(HERE, LIBNAME)

vp = C.c_void_p
i32 = C.c_int
i64 = C.c_long
sz = C.c_size_t
PP = C.POINTER
COMM_ID_BYTES = 128  # ORIGIN_COMM_ID_BYTES

# name -> argtypes  (every function returns int unless listed in _RESTYPE)
SIGNATURES = {
    "origin_last_error": [],
    "origin_abi_version": [],
    "origin_device_count": [PP(i32)],
    "origin_ctx_create": [i32, PP(vp)],
    "origin_ctx_destroy": [vp],
    "origin_sync": [vp],
    "origin_device_name": [vp, C.c_char_p, i32],
    "origin_mem_info": [vp, PP(sz), PP(sz)],
    "origin_stream": [vp, PP(vp)],
    "origin_malloc": [vp, sz, PP(vp)],
    "origin_free": [vp, vp],
    "origin_memset": [vp, vp, i32, sz],
    "origin_h2d": [vp, vp, vp, sz],
    "origin_d2h": [vp, vp, vp, sz],
    "origin_d2h_f32_as_f64": [vp, vp, vp, sz],
    "origin_h2d_f64_as_f32": [vp, vp, vp, sz],
    "origin_d2d": [vp, vp, vp, sz],
    "origin_copy_box": [vp, i32, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32],
    "origin_gather_columns": [vp, vp, i32, i64, vp, i64, i32, vp],
    "origin_scatter_columns": [vp, vp, i32, i64, vp, i64, i32, vp],
    "origin_zmax_map": [vp, vp, vp, i32, i64, vp],
    "origin_count_above": [vp, vp, vp, i32, i64, i32, vp, vp],
    "origin_where_above": [vp, vp, vp, i32, i32, i32, C.c_double, i64, vp, vp, vp, vp, vp, vp],
    "origin_fits_encode": [vp, vp, i32, i64, i32, vp],
    "origin_fits_decode": [vp, vp, i32, i64, i32, vp],
    "origin_fits_write_data": [vp, vp, i32, i64, i32, i32],
    "origin_fits_read_data": [vp, i32, i32, i64, i32, vp],
    "origin_comm_unique_id": [vp],
    "origin_comm_create": [vp, vp, i32, i32, PP(vp)],
    "origin_comm_destroy": [vp],
    "origin_comm_allreduce_f64": [vp, vp, i64],
    "origin_comm_exchange": [vp, i32, vp, vp, vp, i32, vp, vp, vp],
    "origin_timer_start": [vp, i32],
    "origin_timer_stop": [vp, i32],
    "origin_timer_ms": [vp, i32, PP(C.c_float)],
    "origin_prof_enable": [vp, i32],
    "origin_prof_reset": [vp],
    "origin_prof_count": [],
    "origin_prof_get": [vp, i32, PP(C.c_char_p), PP(C.c_double), PP(C.c_long)],
    "origin_dct_fit": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "origin_dct_fit_sums": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp],
    "origin_dct_continuum": [vp, vp, i32, i32, i32, i32, vp],
    "origin_dct_resid_sums": [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp],
    "origin_dct_standardize": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "origin_dct_cont_std": [vp, vp, vp, i32, i32, i32, i32, vp, vp],
    "origin_dct_cont_std_async": [vp, vp, vp, i32, i32, i32, i32, vp, vp],
    "origin_aux_join": [vp],
    "origin_o2": [vp, vp, i32, i64, vp],
    "origin_pca_run": [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, C.c_double, i32, vp, PP(i32),
                       PP(i32), vp, i32],
    "origin_pca_run_into": [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, C.c_double, i32, vp,
                            PP(i32), PP(i32), vp, i32, i32, i64, i64],
    "origin_pca_gram": [vp, vp, vp, vp, i32, i32, vp, vp, vp, i64, vp, vp],
    "origin_pca_eig": [vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp],
    "origin_pca_eig_qrows": [],
    "origin_o2_histogram": [vp, i64, C.c_double, i32, vp, vp, i64, PP(i64), PP(i64)],
    "origin_o2_histogram_batch": [vp, vp, i32, C.c_double, i32, vp, vp, i64, vp],
    "origin_gauss_fit": [vp, vp, i64, vp, PP(i32), PP(i32)],
    "origin_o2_threshold_batch": [vp, vp, vp, i32, i64, C.c_double, vp, vp],
    "origin_o2_areas_fit": [vp, vp, vp, i32, C.c_double, i32, C.c_double, vp, vp, vp, i64, vp, vp, vp],
    "origin_glr_plan_create": [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, PP(vp)],
    "origin_glr_plan_destroy": [vp],
    "origin_glr_plan_set_precision": [vp, i32],
    "origin_glr_plan_get_precision": [vp, PP(i32)],
    "origin_glr_plan_bytes": [vp, PP(sz)],
    "origin_glr_plan_mfma_count": [vp, PP(i64), PP(i64)],
    "origin_glr_plan_fold_eps": [vp, PP(C.c_float), PP(i32)],
    "origin_glr_mfma_count_model": [i32, i32, i32, i32, i32, i32, i32, i32, PP(i64), PP(i64)],
    "origin_glr_work_elems": [vp, PP(sz)],
    "origin_glr_run": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "origin_glr_rows_supported": [vp, PP(i32)],
    "origin_glr_run_rows": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32],
    "origin_glr_run_rect": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32],
    "origin_glr_run_finish": [vp, vp, vp, vp, vp],
    "origin_pca_set_tail_hook": [vp, vp, vp, i32],
    "origin_local_max": [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp],
    "origin_local_max_sparse_plan": [vp, i32, i32, i32, PP(i64), PP(i32)],
    "origin_local_max_sparse": [vp, vp, vp, vp, i32, i32, i32, i64, i32, vp, vp, vp, vp, vp],
    "origin_sparse_to_dense": [vp, vp, vp, vp, i64, i32, vp, i64],
    "origin_sparse_count_above": [vp, vp, vp, vp, i64, i32, vp, i64, i32, vp, vp],
    "origin_sparse_where_above": [vp, vp, vp, vp, i64, i32, C.c_double, vp, i64, vp, vp, vp,
                                  PP(i64)],
    "origin_sparse_zmax_map": [vp, vp, vp, vp, i64, i32, vp, i64, vp],
}
_RESTYPE = {"origin_last_error": C.c_char_p}

ERROR_NAMES = {-1: "ORIGIN_E_ARG", -2: "ORIGIN_E_NOMEM", -3: "ORIGIN_E_HIP",
               -4: "ORIGIN_E_NODEVICE", -5: "ORIGIN_E_STATE"}

_lib = None


class OriginHipError(RuntimeError):
    """A liborigin_hip call failed (Step.__call__ turns it into Status.FAILED,
    reference steps.py:272-278)."""

    def __init__(self, code, msg):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {msg}")
        self.code = code


def load():
    """Load the shared library and declare every prototype; raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise RuntimeError(
            f"{LIBPATH} is missing: build it with `python -m origin_amd.build` "
            "(hipcc --offload-arch=gfx950).  origin_amd has no CPU fallback.")
    lib = C.CDLL(LIBPATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, i32)
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load().origin_last_error()
        raise OriginHipError(code, msg.decode() if msg else "unknown error")


def call(name, *args):
    check(getattr(load(), name)(*args))
