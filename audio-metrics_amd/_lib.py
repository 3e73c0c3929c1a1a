"""ctypes binding of include/audio_metrics_hip.h (the C-ABI drop-in boundary).

The product path has NO fallback: if the shared object is missing or a symbol
cannot be resolved this module raises, it never routes to a CPU implementation.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p

from ._build import DEV_LIB_PATH, LIB_PATH

_P = c_void_p

# name -> (restype, argtypes); mirrors include/audio_metrics_hip.h one to one
SIGNATURES = {
    "am_version": (c_char_p, []),
    "am_status_string": (c_char_p, [c_int]),
    "am_last_error": (c_char_p, []),
    "am_stats_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "am_stats_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "am_colsum_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, c_size_t, _P]),
    "am_scatter_f32": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "am_stats_f64_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "am_stats_f64": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "am_colsum_f64": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, c_size_t, _P]),
    "am_scatter_f64": (c_int, [_P, c_int64, c_int, c_int64, _P, _P, _P, c_size_t, _P]),
    "am_stats_merge_f64": (c_int, [c_int64, _P, _P, c_int64, _P, _P, c_int, _P, _P, _P]),
    "am_stats_push_max_rows": (c_int, []),
    "am_stats_push_f32": (c_int, [_P, c_int64, c_int, c_int64, c_int64, _P, _P, _P, _P, c_int64, _P]),
    "am_frechet_workspace_bytes": (c_size_t, [c_int]),
    "am_frechet_f64": (c_int, [_P, _P, _P, _P, c_int, c_int, c_double, _P, _P, c_size_t, _P]),
    "am_frechet_first_block": (c_int, []),
    "am_frechet_enqueue_f64": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_double, _P, _P, c_size_t, _P]),
    "am_apa_f64": (c_double, [c_double, c_double, c_double]),
    "am_kd_workspace_bytes": (c_size_t, [c_int, c_int]),
    "am_kd_poly_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "am_kd_poly_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int, _P, _P, c_int, c_int,
                               c_double, c_double, c_int, _P, _P, c_size_t, _P]),
    "am_kd_draw_indices": (c_int, [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, c_int64, c_int64, c_int,
                                   c_int, _P, _P]),
    "am_kd_rbf_workspace_bytes": (c_size_t, [c_int, c_int]),
    "am_kd_rbf_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int, _P, _P, c_int, c_int, c_double, _P, _P,
                              c_size_t, _P]),
    "am_knn_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "am_knn_radii_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int, c_int, _P, _P, c_size_t, _P]),
    "am_knn_sym_eligible": (c_int, [c_int64, c_int, c_int]),
    "am_knn_list_width": (c_int, [c_int]),
    "am_knn_part_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "am_knn_bounds_f32": (c_int, [_P, c_int64, c_int64, c_int, c_int, c_int64, c_int64, _P, _P, c_size_t, _P]),
    "am_knn_sym_part_f32": (c_int, [_P, c_int64, c_int64, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "am_knn_lists_finish_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "am_knn_lists_finish_f32": (c_int, [_P, c_int, _P, c_int64, c_int64, c_int, c_int, _P, _P, c_size_t, _P]),
    "am_prdc_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "am_prdc_counts_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int, _P, _P, _P, _P, _P, _P,
                                   _P, c_size_t, _P]),
    "am_prdc_reduce": (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P]),
    "am_prepared_half_ld": (c_int64, [c_int]),
    "am_prepare_set_f32": (c_int, [_P, c_int64, c_int64, c_int, _P, _P, _P, _P]),
    "am_knn_radii_prepared_f32": (c_int, [_P, c_int64, c_int64, c_int, _P, c_int, _P, _P, c_size_t, _P]),
    "am_knn_bounds_prepared_f32": (c_int, [_P, c_int64, c_int64, c_int, _P, c_int, c_int64, c_int64, _P, _P, c_size_t, _P]),
    "am_knn_sym_part_prepared_f32": (c_int, [_P, c_int64, c_int64, c_int, _P, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "am_prdc_counts_prepared_f32": (c_int, [_P, c_int64, c_int64, _P, _P, c_int64, c_int64, _P, c_int, _P, _P, _P, _P, _P, _P,
                                            _P, c_size_t, _P]),
    "am_knn_f64_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "am_knn_radii_f64": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int, c_int, _P, _P, c_size_t, _P]),
    "am_prdc_f64_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "am_prdc_counts_f64": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int, _P, _P, _P, _P, _P, _P,
                                   _P, c_size_t, _P]),
    "am_kd_f64_workspace_bytes": (c_size_t, [c_int, c_int]),
    "am_kd_poly_f64": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int, _P, _P, c_int, c_int,
                               c_double, c_double, c_int, _P, _P, c_size_t, _P]),
    "am_kd_rbf_f64": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int, _P, _P, c_int, c_int, c_double, _P, _P,
                              c_size_t, _P]),
    "am_eigh_workspace_bytes": (c_size_t, [c_int]),
    "am_eigh_sym_f64": (c_int, [_P, c_int, _P, _P, c_int, _P, c_size_t, _P]),
    "am_project_f64": (c_int, [_P, c_int64, c_int64, c_int, _P, _P, c_int, _P, _P]),
    "am_project_rows_f64": (c_int, [_P, c_int64, c_int64, c_int, _P, _P, c_int, _P, _P]),
    "am_evaluate_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int, c_int, c_int, ctypes.c_uint]),
    "am_evaluate_f32": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int, ctypes.c_uint, c_int, _P, _P, c_int, c_int,
                                c_double, c_double, c_int, _P, _P, _P, _P, c_size_t, _P, _P]),
    "am_evaluate_sharded_workspace_bytes": (c_size_t, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.c_uint]),
    "am_evaluate_sharded_f32": (c_int, [_P, c_int64, _P, c_int64, c_int, _P, _P, _P, ctypes.c_uint, c_int, _P, _P, c_int, c_int,
                                        c_double, c_double, c_int, _P, _P, c_size_t, _P, _P, _P]),
    "am_kernel_clock_enable": (c_int, [c_int]),
    "am_kernel_clock_read": (c_int, [c_int, _P, _P]),
    "am_knn_path": (c_int, [c_int64, c_int64, c_int, c_int, c_int]),
    "am_prdc_path": (c_int, [c_int64, c_int64, c_int]),
    "am_filter_engine": (c_int, [c_int]),
    "am_filter_stats_enable": (c_int, [_P]),
}


class HipLibraryError(RuntimeError):
    pass


class PreparedSetStruct(ctypes.Structure):
    """am_prepared_set: a host struct of three device pointers."""
    _fields_ = [("norms", c_void_p), ("stats", c_void_p), ("half", c_void_p)]


class EvaluateSideStruct(ctypes.Structure):
    """am_evaluate_side: results the caller already holds / wants to keep (device pointers, NULL = not given)."""
    _fields_ = [("mean", c_void_p), ("cov", c_void_p), ("radii", c_void_p),
                ("mean_out", c_void_p), ("cov_out", c_void_p), ("radii_out", c_void_p)]


# am_collectives: the two collectives of am_evaluate_sharded_f32 as hooks (include/audio_metrics_hip.h)
ALL_REDUCE_SUM_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p)
ALL_GATHER_V_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64), c_void_p)


class CollectivesStruct(ctypes.Structure):
    _fields_ = [("ctx", c_void_p), ("rank", c_int), ("world", c_int),
                ("all_reduce_sum", ALL_REDUCE_SUM_FN), ("all_gather_v", ALL_GATHER_V_FN)]


_lib = None


def library_path():
    """The shipped library; AM_HIP_LIBRARY=dev (tools and fallback-path tests only) selects the -DAM_DEV_KNOBS build."""
    choice = os.environ.get("AM_HIP_LIBRARY", "")
    if choice == "dev":
        return DEV_LIB_PATH
    if choice.endswith(".so"):                            # an experimental variant build (tools/ only)
        return choice if os.path.isabs(choice) else os.path.join(os.path.dirname(LIB_PATH), choice)
    return LIB_PATH


def load():
    """Load libaudio_metrics_hip.so (built in-tree by ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise HipLibraryError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python __graft_entry__.py build`). There is no CPU fallback.")
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        lib = load()
        msg = lib.am_last_error().decode() or lib.am_status_string(status).decode()
        raise HipLibraryError(f"{what} failed with status {status}: {msg}")
