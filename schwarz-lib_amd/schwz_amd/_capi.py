"""ctypes binding of libschwz_hip.so (the C ABI declared in include/schwz_hip.h).

The library is built in-tree by `make -C schwarz-lib_amd` (see
__graft_entry__.build).  There is no fallback: if the shared object is missing
the import of this module raises.
"""
import ctypes as C
import os

import numpy as np

# torch bundles its own libamdhip64.so.7; libschwz_hip.so links the same SONAME.
# Importing torch FIRST makes the dynamic linker hand that one HIP runtime to both,
# otherwise two runtimes initialise in one process and the second sees no device.
import torch  # noqa: F401,E402

_PKG = os.path.dirname(os.path.abspath(__file__))
# SCHWZ_HIP_LIB: another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("SCHWZ_HIP_LIB") or os.path.join(os.path.dirname(_PKG), "lib", "libschwz_hip.so")

OK = 0
ERR_INVALID, ERR_HIP, ERR_NOT_IMPLEMENTED, ERR_NOT_SPD, ERR_IO, ERR_DIVERGED = 1, 2, 3, 4, 5, 6
OP_ADD, OP_COPY, OP_DIFF, OP_AVG = 0, 1, 2, 3
SOLVER_ITERATIVE, SOLVER_DIRECT = 0, 1
PRECOND_NONE, PRECOND_JACOBI, PRECOND_BLOCK_JACOBI, PRECOND_ILU, PRECOND_ISAI = 0, 1, 2, 3, 4

# every symbol include/schwz_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "schwz_last_error", "schwz_version", "schwz_device_count", "schwz_set_device", "schwz_setup_threads",
    "schwz_gather", "schwz_scatter", "schwz_gather_typed", "schwz_scatter_typed",
    "schwz_csr_create", "schwz_csr_destroy", "schwz_csr_nnz", "schwz_csr_format", "schwz_csr_symmetric", "schwz_csr_matrix_bytes", "schwz_csr_sweep_slots", "schwz_csr_sweep_left_out", "schwz_csr_spmv",
    "schwz_pcg_create", "schwz_pcg_create_ex", "schwz_pcg_destroy", "schwz_pcg_flavour", "schwz_pcg_solve",
    "schwz_gmres_create", "schwz_gmres_destroy", "schwz_gmres_solve", "schwz_gmres_last_stats",
    "schwz_profile_begin", "schwz_profile_end", "schwz_profile_kind", "schwz_stream_probe",
    "schwz_trs_create", "schwz_trs_destroy", "schwz_trs_solve",
    "schwz_problem_laplacian", "schwz_problem_from_csr", "schwz_problem_from_matrix_market",
    "schwz_problem_destroy", "schwz_problem_size", "schwz_problem_nnz", "schwz_problem_row",
    "schwz_problem_permute",
    "schwz_rhs_random", "schwz_partition_regular", "schwz_partition_regular2d", "schwz_partition_graph",
    "schwz_subdomain_setup", "schwz_subdomain_destroy", "schwz_subdomain_sizes",
    "schwz_subdomain_local_to_global", "schwz_subdomain_local_matrix",
    "schwz_subdomain_interface_matrix", "schwz_subdomain_get_list",
    "schwz_subdomain_add_put_list", "schwz_subdomain_put_list",
    "schwz_subdomain_send_offset", "schwz_subdomain_recv_offset",
    "schwz_problem_from_rows", "schwz_problem_extract_rows",
    "schwz_cholesky", "schwz_ilu0", "schwz_isai", "schwz_free",
    "schwz_subdomain_to_device", "schwz_ras_pack", "schwz_ras_unpack", "schwz_ras_pack_f32",
    "schwz_ras_unpack_f32", "schwz_ras_pack_neighbor", "schwz_ras_unpack_neighbor",
    "schwz_ras_early_pack_ok", "schwz_ras_pack_early",
    "schwz_window_alloc", "schwz_window_free", "schwz_window_export", "schwz_window_open", "schwz_window_close",
    "schwz_host_atomic_add_i32", "schwz_host_atomic_load_i32", "schwz_host_atomic_store_i32",
    "schwz_host_atomic_min_f64",
    "schwz_ras_update_boundary", "schwz_ras_local_residual", "schwz_ras_local_residual_launch",
    "schwz_ras_local_residual_wait", "schwz_ras_local_solve", "schwz_ras_set_local_max_iters",
    "schwz_ras_last_inner_stats",
    "schwz_ras_check_and_solve_launch",
    "schwz_ras_norm_sq_to_device",
    "schwz_ras_restrict", "schwz_ras_vector", "schwz_ras_local_csr", "schwz_ras_jacobi_form", "schwz_ras_cg_flavour", "schwz_ras_get_interior",
    "schwz_ras_true_residual_sq", "schwz_ras_algorithmic_bytes",
]


class SchwzError(RuntimeError):
    """Mirrors the reference's Error hierarchy (include/exception.hpp:42-210)."""

    def __init__(self, code, msg):
        super().__init__("schwz error %d: %s" % (code, msg))
        self.code = code


class NotImplementedSchwz(SchwzError):
    pass


class SolverOptions(C.Structure):
    _fields_ = [
        ("local_solver", C.c_int32),
        ("precond", C.c_int32),
        ("local_tol", C.c_double),
        ("local_max_iters", C.c_int32),
        ("natural_factor_ordering", C.c_int32),
        ("spmv_variant", C.c_int32),
        ("precond_block_size", C.c_int32),
        ("non_symmetric", C.c_int32),
        ("restart_iter", C.c_int32),
    ]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libschwz_hip.so not found at %s: build it with `make -C schwarz-lib_amd` "
        "(there is no CPU fallback)" % LIB_PATH)

lib = C.CDLL(LIB_PATH)

vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
pvp = C.POINTER(C.c_void_p)


def _sig(name, restype, argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = argtypes


_sig("schwz_last_error", C.c_char_p, [])
_sig("schwz_version", C.c_char_p, [])
_sig("schwz_device_count", i32, [])
_sig("schwz_setup_threads", i32, [])
_sig("schwz_set_device", i32, [i32])
_sig("schwz_gather", i32, [i64, vp, vp, vp, i32, vp])
_sig("schwz_scatter", i32, [i64, vp, vp, vp, i32, vp])
_sig("schwz_gather_typed", i32, [i64, vp, i32, vp, vp, i32, i32, vp])
_sig("schwz_scatter_typed", i32, [i64, vp, i32, vp, vp, i32, i32, vp])
_sig("schwz_csr_create", i32, [i64, i64, vp, vp, vp, pvp])
_sig("schwz_csr_destroy", None, [vp])
_sig("schwz_csr_nnz", i64, [vp])
_sig("schwz_csr_format", i32, [vp])
_sig("schwz_csr_symmetric", i32, [vp])
_sig("schwz_csr_matrix_bytes", i64, [vp, i32])
_sig("schwz_csr_sweep_slots", i32, [vp])
_sig("schwz_csr_sweep_left_out", i32, [vp])
_sig("schwz_csr_spmv", i32, [vp, dbl, vp, dbl, vp, i32, vp])
_sig("schwz_pcg_create", i32, [vp, i32, pvp])
_sig("schwz_pcg_create_ex", i32, [vp, i32, i32, pvp])
_sig("schwz_pcg_destroy", None, [vp])
_sig("schwz_pcg_flavour", i32, [vp])
_sig("schwz_gmres_create", i32, [vp, i32, i32, i32, pvp])
_sig("schwz_gmres_destroy", None, [vp])
_sig("schwz_gmres_last_stats", i32, [vp, C.POINTER(C.c_int), C.POINTER(dbl)])
_sig("schwz_gmres_solve", i32, [vp, vp, vp, dbl, i32, C.POINTER(C.c_int), C.POINTER(dbl), vp])
_sig("schwz_pcg_solve", i32, [vp, vp, vp, dbl, i32, C.POINTER(C.c_int), C.POINTER(dbl), vp])
_sig("schwz_profile_begin", i32, [i32])
_sig("schwz_profile_end", i32, [C.POINTER(dbl), C.POINTER(i64)])
_sig("schwz_profile_kind", i32, [i32, C.POINTER(dbl), C.POINTER(i64)])
_sig("schwz_stream_probe", i32, [i64, i32, vp, vp, vp])
_sig("schwz_trs_create", i32, [i64] + [vp] * 7 + [pvp])
_sig("schwz_trs_destroy", None, [vp])
_sig("schwz_trs_solve", i32, [vp, vp, vp, vp])
_sig("schwz_problem_laplacian", i32, [i32, i64, i64, i64, pvp])
_sig("schwz_problem_from_csr", i32, [i64, vp, vp, vp, pvp])
_sig("schwz_problem_from_rows", i32, [i64, i64, vp, vp, vp, vp, pvp])
_sig("schwz_problem_extract_rows", i32, [vp, i64, vp, vp, vp, vp])
_sig("schwz_problem_from_matrix_market", i32, [C.c_char_p, pvp])
_sig("schwz_problem_destroy", None, [vp])
_sig("schwz_problem_size", i64, [vp])
_sig("schwz_problem_nnz", i64, [vp])
_sig("schwz_problem_row", i32, [vp, i64, C.POINTER(C.c_int), vp, vp, i32])
_sig("schwz_problem_permute", i32, [vp, i32, vp, vp, vp, pvp])
_sig("schwz_rhs_random", i32, [i64, vp, vp])
_sig("schwz_partition_regular", i32, [i64, i32, vp])
_sig("schwz_partition_regular2d", i32, [i64, i32, vp])
_sig("schwz_partition_graph", i32, [vp, i32, vp])
_sig("schwz_subdomain_setup", i32, [vp, i32, i32, i32, vp, pvp])
_sig("schwz_subdomain_destroy", None, [vp])
_sig("schwz_subdomain_sizes", i32, [vp, vp])
_sig("schwz_subdomain_local_to_global", i32, [vp, vp])
_sig("schwz_subdomain_local_matrix", i32, [vp, vp, vp, vp])
_sig("schwz_subdomain_interface_matrix", i32, [vp, vp, vp, vp])
_sig("schwz_subdomain_get_list", i32, [vp, i32, C.POINTER(C.c_int), C.POINTER(i64), vp])
_sig("schwz_subdomain_put_list", i32, [vp, i32, C.POINTER(C.c_int), C.POINTER(i64), vp])
_sig("schwz_subdomain_add_put_list", i32, [vp, i32, i64, vp])
_sig("schwz_subdomain_send_offset", i32, [vp, i32, C.POINTER(i64)])
_sig("schwz_subdomain_recv_offset", i32, [vp, i32, C.POINTER(i64)])
_sig("schwz_cholesky", i32, [i64, vp, vp, vp, i32] + [pvp] * 7)
_sig("schwz_ilu0", i32, [i64, vp, vp, vp] + [pvp] * 6)
_sig("schwz_isai", i32, [i64, vp, vp, vp, i32, pvp])
_sig("schwz_free", None, [vp])
_sig("schwz_subdomain_to_device", i32, [vp, vp, C.POINTER(SolverOptions)])
_sig("schwz_ras_pack", i32, [vp, vp, vp])
_sig("schwz_ras_unpack", i32, [vp, vp, vp])
_sig("schwz_ras_pack_f32", i32, [vp, vp, vp])
_sig("schwz_ras_unpack_f32", i32, [vp, vp, vp])
_sig("schwz_ras_update_boundary", i32, [vp, vp])
_sig("schwz_ras_pack_neighbor", i32, [vp, i32, vp, i32, vp])
_sig("schwz_ras_early_pack_ok", i32, [vp])
_sig("schwz_ras_pack_early", i32, [vp, vp, i32, vp])
_sig("schwz_ras_unpack_neighbor", i32, [vp, i32, vp, i32, vp])
_sig("schwz_window_alloc", i32, [i64, pvp])
_sig("schwz_window_free", i32, [vp])
_sig("schwz_window_export", i32, [vp, vp])
_sig("schwz_window_open", i32, [vp, pvp])
_sig("schwz_window_close", i32, [vp])
_sig("schwz_host_atomic_add_i32", i32, [vp, i32])
_sig("schwz_host_atomic_load_i32", i32, [vp])
_sig("schwz_host_atomic_store_i32", None, [vp, i32])
_sig("schwz_host_atomic_min_f64", dbl, [vp, dbl])
_sig("schwz_ras_local_residual", i32, [vp, C.POINTER(dbl), vp])
_sig("schwz_ras_local_residual_launch", i32, [vp, vp])
_sig("schwz_ras_local_residual_wait", i32, [vp, C.POINTER(dbl)])
_sig("schwz_ras_check_and_solve_launch", i32, [vp, vp])
_sig("schwz_ras_norm_sq_to_device", i32, [vp, vp, vp])
_sig("schwz_ras_local_solve", i32, [vp, C.POINTER(C.c_int), vp])
_sig("schwz_ras_set_local_max_iters", i32, [vp, i32])
_sig("schwz_ras_last_inner_stats", i32, [vp, C.POINTER(C.c_int), C.POINTER(dbl)])
_sig("schwz_ras_restrict", i32, [vp, vp])
_sig("schwz_ras_vector", i32, [vp, i32, pvp, C.POINTER(i64)])
_sig("schwz_ras_local_csr", i32, [vp, pvp])
_sig("schwz_ras_jacobi_form", i32, [vp])
_sig("schwz_ras_cg_flavour", i32, [vp])
_sig("schwz_ras_get_interior", i32, [vp, vp, vp])
_sig("schwz_ras_true_residual_sq", i32, [vp, C.POINTER(dbl), vp])
_sig("schwz_ras_algorithmic_bytes", i64, [vp, i32])


def check(rc):
    if rc != OK:
        msg = lib.schwz_last_error().decode("utf-8", "replace")
        if rc == ERR_NOT_IMPLEMENTED:
            raise NotImplementedSchwz(rc, msg)
        raise SchwzError(rc, msg)


def ptr(a):
    """void* of a numpy array (host) or a raw integer address (device)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(int(a))


def device_count():
    return int(lib.schwz_device_count())
