"""ctypes binding of the CPU oracle (oracle/libschwz_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libschwz_oracle.so")

IDX = np.int32
c_idx_p = C.POINTER(C.c_int32)
c_dbl_p = C.POINTER(C.c_double)

OP_ADD, OP_COPY, OP_DIFF, OP_AVG = 0, 1, 2, 3
SOLVER_ITERATIVE, SOLVER_DIRECT = 0, 1
PRECOND_NONE, PRECOND_JACOBI, PRECOND_BLOCK_JACOBI, PRECOND_ILU = 0, 1, 2, 3


class Settings(C.Structure):
    _fields_ = [
        ("max_iters", C.c_int32),
        ("tol", C.c_double),
        ("overlap", C.c_int32),
        ("local_solver", C.c_int32),
        ("precond", C.c_int32),
        ("local_tol", C.c_double),
        ("local_max_iters", C.c_int32),
        ("enable_global_check", C.c_int32),
        ("enable_onesided", C.c_int32),
        ("global_check_iter_offset", C.c_int32),
        ("natural_factor_ordering", C.c_int32),
        ("num_threads", C.c_int32),
        ("enable_overlap", C.c_int32),
        ("use_mixed_precision", C.c_int32),
        ("precond_block_size", C.c_int32),
        ("reset_local_crit_iter", C.c_int32),
        ("updated_max_iters", C.c_int32),
        ("non_symmetric", C.c_int32),
        ("restart_iter", C.c_int32),
    ]


def make_settings(max_iters=100, tol=1e-6, overlap=2, local_solver=SOLVER_ITERATIVE,
                  precond=PRECOND_NONE, local_tol=1e-12, local_max_iters=-1,
                  enable_global_check=1, enable_onesided=0, global_check_iter_offset=0,
                  natural_factor_ordering=0, num_threads=0, enable_overlap=0, use_mixed_precision=0,
                  precond_block_size=1, reset_local_crit_iter=-1, updated_max_iters=-1,
                  non_symmetric=0, restart_iter=1):
    """Defaults follow benchmarking/bench_base.hpp:50-144 except
    enable_global_check (needed to ever stop)."""
    return Settings(max_iters, tol, overlap, local_solver, precond, local_tol,
                    local_max_iters, enable_global_check, enable_onesided,
                    global_check_iter_offset, natural_factor_ordering, num_threads, enable_overlap,
                    use_mixed_precision, precond_block_size, reset_local_crit_iter, updated_max_iters,
                    int(non_symmetric), restart_iter)


class Result(C.Structure):
    _fields_ = [
        ("iter_count", C.c_int32),
        ("converged", C.c_int32),
        ("residual_norm", C.c_double),
        ("rhs_norm", C.c_double),
        ("sol_norm", C.c_double),
        ("elapsed_s", C.c_double),
        ("setup_s", C.c_double),
    ]


def build(force=False):
    if force or not os.path.exists(_SO) or (
            os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "schwz_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    vp = C.c_void_p
    i64 = C.c_int64
    L.schwz_or_laplacian2d.restype = i64
    L.schwz_or_laplacian2d.argtypes = [C.c_int, vp, vp, vp]
    L.schwz_or_laplacian3d.restype = i64
    L.schwz_or_laplacian3d.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.schwz_or_rhs_random.argtypes = [i64, vp]
    L.schwz_or_first_rows_regular.argtypes = [i64, C.c_int, vp]
    L.schwz_or_partition_regular2d.argtypes = [C.c_int, C.c_int, vp]
    L.schwz_or_partition_regular2d.restype = C.c_int
    L.schwz_or_apply_partition.argtypes = [i64, C.c_int] + [vp] * 10
    L.schwz_or_subdomain_setup.restype = vp
    L.schwz_or_subdomain_setup.argtypes = [i64, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]
    L.schwz_or_subdomain_free.argtypes = [vp]
    L.schwz_or_subdomain_sizes.argtypes = [vp, vp]
    for name in ("local_to_global", "local_rp", "local_col", "iface_rp", "iface_col"):
        f = getattr(L, "schwz_or_sd_" + name)
        f.restype = c_idx_p
        f.argtypes = [vp]
    for name in ("local_val", "iface_val"):
        f = getattr(L, "schwz_or_sd_" + name)
        f.restype = c_dbl_p
        f.argtypes = [vp]
    for name in ("get_list", "put_list"):
        f = getattr(L, "schwz_or_sd_" + name)
        f.restype = C.c_int
        f.argtypes = [vp, C.c_int, C.POINTER(C.c_int32), C.POINTER(c_idx_p)]
    L.schwz_or_sd_add_put_list.argtypes = [vp, C.c_int, C.c_int32, vp]
    L.schwz_or_state_create.restype = vp
    L.schwz_or_state_create.argtypes = [vp, vp, C.POINTER(Settings)]
    L.schwz_or_state_free.argtypes = [vp]
    L.schwz_or_state_global_solution.restype = c_dbl_p
    L.schwz_or_state_global_solution.argtypes = [vp]
    L.schwz_or_state_local_solution.restype = c_dbl_p
    L.schwz_or_state_local_solution.argtypes = [vp]
    L.schwz_or_state_local_rhs.restype = c_dbl_p
    L.schwz_or_state_local_rhs.argtypes = [vp]
    L.schwz_or_state_factors.restype = C.c_int
    L.schwz_or_state_factors.argtypes = [vp] + [C.POINTER(vp)] * 7
    L.schwz_or_pack.argtypes = [vp, C.c_int, vp]
    L.schwz_or_unpack.argtypes = [vp, C.c_int, vp]
    L.schwz_or_update_boundary.argtypes = [vp]
    L.schwz_or_local_residual.restype = C.c_double
    L.schwz_or_local_residual.argtypes = [vp]
    L.schwz_or_local_solve.restype = C.c_int
    L.schwz_or_local_solve.argtypes = [vp]
    L.schwz_or_restrict.argtypes = [vp]
    L.schwz_or_state_set_local_max_iters.argtypes = [vp, C.c_int32]
    L.schwz_or_state_set_local_max_iters.restype = None
    L.schwz_or_ras_run.restype = C.c_int
    L.schwz_or_ras_run.argtypes = [i64, vp, vp, vp, vp, C.c_int, vp, C.POINTER(Settings),
                                   vp, vp, vp, vp, C.POINTER(Result)]
    L.schwz_or_spmv.argtypes = [i64, vp, vp, vp, C.c_double, vp, C.c_double, vp]
    L.schwz_or_gather.argtypes = [i64, vp, vp, vp, C.c_int]
    L.schwz_or_scatter.argtypes = [i64, vp, vp, vp, C.c_int]
    L.schwz_or_pcg.restype = C.c_int
    L.schwz_or_pcg.argtypes = [i64, vp, vp, vp, vp, vp, C.c_int, C.c_double, C.c_int, vp]
    L.schwz_or_pcg_ex.restype = C.c_int
    L.schwz_or_pcg_ex.argtypes = [i64, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_int, vp]
    L.schwz_or_gmres.restype = C.c_int
    L.schwz_or_gmres.argtypes = [i64, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, vp]
    L.schwz_or_jacobi_blocks.argtypes = [i64, vp, vp, C.c_int, vp]
    L.schwz_or_jacobi_blocks.restype = i64
    L.schwz_or_ilu0.argtypes = [i64, vp, vp, vp] + [C.POINTER(vp)] * 6
    L.schwz_or_isai.argtypes = [i64, vp, vp, vp, C.c_int, C.POINTER(vp)]
    L.schwz_or_isai.restype = None
    L.schwz_or_cholesky.restype = C.c_int
    L.schwz_or_cholesky.argtypes = [i64, vp, vp, vp, C.c_int] + [C.POINTER(vp)] * 7
    L.schwz_or_direct_solve.argtypes = [i64] + [vp] * 10
    L.schwz_or_free.argtypes = [vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _np_from(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).copy()


# --------------------------------------------------------------------------
# generators / partition
# --------------------------------------------------------------------------

def laplacian2d(n):
    L = lib()
    N = n * n
    rp = np.zeros(N + 1, dtype=IDX)
    nnz = L.schwz_or_laplacian2d(n, _p(rp), None, None)
    col = np.zeros(nnz, dtype=IDX)
    val = np.zeros(nnz, dtype=np.float64)
    L.schwz_or_laplacian2d(n, _p(rp), _p(col), _p(val))
    return rp, col, val


def laplacian3d(nx, ny=None, nz=None):
    L = lib()
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    N = nx * ny * nz
    rp = np.zeros(N + 1, dtype=IDX)
    nnz = L.schwz_or_laplacian3d(nx, ny, nz, _p(rp), None, None)
    col = np.zeros(nnz, dtype=IDX)
    val = np.zeros(nnz, dtype=np.float64)
    L.schwz_or_laplacian3d(nx, ny, nz, _p(rp), _p(col), _p(val))
    return rp, col, val


def rhs_random(N):
    out = np.zeros(N, dtype=np.float64)
    lib().schwz_or_rhs_random(N, _p(out))
    return out


def first_rows_regular(N, P):
    fr = np.zeros(P + 1, dtype=IDX)
    lib().schwz_or_first_rows_regular(N, P, _p(fr))
    return fr


def partition_regular2d(n1d, P):
    part = np.zeros(n1d * n1d, dtype=np.uint32)
    rc = lib().schwz_or_partition_regular2d(n1d, P, _p(part))
    if rc != 0:
        raise ValueError("regular2d needs a square subdomain count dividing the grid")
    return part


def apply_partition(rp, col, val, part, P):
    N = len(rp) - 1
    perm = np.zeros(N, dtype=IDX)
    iperm = np.zeros(N, dtype=IDX)
    fr = np.zeros(P + 1, dtype=IDX)
    orp = np.zeros(N + 1, dtype=IDX)
    ocol = np.zeros(len(col), dtype=IDX)
    oval = np.zeros(len(val), dtype=np.float64)
    part = np.ascontiguousarray(part, dtype=np.uint32)
    lib().schwz_or_apply_partition(N, P, _p(part), _p(rp), _p(col), _p(val), _p(perm),
                                   _p(iperm), _p(fr), _p(orp), _p(ocol), _p(oval))
    return perm, iperm, fr, orp, ocol, oval


# --------------------------------------------------------------------------
# subdomain
# --------------------------------------------------------------------------

class Subdomain:
    """Index sets and matrices of one subdomain (restricted_schwarz.cpp:56-604)."""

    def __init__(self, rp, col, val, P, me, overlap, first_row):
        self._L = lib()
        self._keep = (rp, col, val, first_row)
        self.N = len(rp) - 1
        self.P, self.me = P, me
        self.h = self._L.schwz_or_subdomain_setup(self.N, _p(rp), _p(col), _p(val), P, me,
                                                  overlap, _p(first_row))
        self._sizes()

    def _sizes(self):
        s = np.zeros(10, dtype=np.int64)
        self._L.schwz_or_subdomain_sizes(self.h, _p(s))
        (self.local_size, self.local_size_x, self.overlap_size, self.halo_size,
         self.nnz_local, self.nnz_interface, self.num_neighbors_in, self.num_neighbors_out,
         self.num_recv, self.num_send) = [int(v) for v in s]

    @property
    def local_to_global(self):
        return _np_from(self._L.schwz_or_sd_local_to_global(self.h),
                        self.local_size_x + self.halo_size, IDX)

    def local_matrix(self):
        n = self.local_size_x
        return (_np_from(self._L.schwz_or_sd_local_rp(self.h), n + 1, IDX),
                _np_from(self._L.schwz_or_sd_local_col(self.h), self.nnz_local, IDX),
                _np_from(self._L.schwz_or_sd_local_val(self.h), self.nnz_local, np.float64))

    def interface_matrix(self):
        n = self.local_size_x
        return (_np_from(self._L.schwz_or_sd_iface_rp(self.h), n + 1, IDX),
                _np_from(self._L.schwz_or_sd_iface_col(self.h), self.nnz_interface, IDX),
                _np_from(self._L.schwz_or_sd_iface_val(self.h), self.nnz_interface,
                         np.float64))

    def _list(self, fn, k):
        cnt = C.c_int32(0)
        ids = c_idx_p()
        rank = fn(self.h, k, C.byref(cnt), C.byref(ids))
        return rank, _np_from(ids, cnt.value, IDX)

    def get_lists(self):
        return [self._list(self._L.schwz_or_sd_get_list, k)
                for k in range(self.num_neighbors_in)]

    def put_lists(self):
        return [self._list(self._L.schwz_or_sd_put_list, k)
                for k in range(self.num_neighbors_out)]

    def add_put_list(self, p, ids):
        ids = np.ascontiguousarray(ids, dtype=IDX)
        self._L.schwz_or_sd_add_put_list(self.h, p, len(ids), _p(ids))
        self._sizes()

    def close(self):
        if self.h:
            self._L.schwz_or_subdomain_free(self.h)
            self.h = None


def connect(subdomains):
    """The index handshake of restricted_schwarz.cpp:400-472 for in-process
    subdomains: q's get list for p becomes p's put list for q."""
    P = len(subdomains)
    for p in range(P):
        for q in range(P):
            if q == p:
                continue
            for rank, ids in subdomains[q].get_lists():
                if rank == p:
                    subdomains[p].add_put_list(q, ids)


class State:
    """Per-subdomain iteration state and the five loop steps (A.3)."""

    def __init__(self, sd, rhs, settings):
        self._L = lib()
        self.sd = sd
        self._rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        self._s = settings
        self.h = self._L.schwz_or_state_create(sd.h, _p(self._rhs), C.byref(settings))

    def global_solution(self):
        return np.ctypeslib.as_array(self._L.schwz_or_state_global_solution(self.h),
                                     shape=(self.sd.N,))

    def local_solution(self):
        return np.ctypeslib.as_array(self._L.schwz_or_state_local_solution(self.h),
                                     shape=(self.sd.local_size_x,))

    def local_rhs(self):
        return _np_from(self._L.schwz_or_state_local_rhs(self.h), self.sd.local_size_x,
                        np.float64)

    def pack(self, k):
        cnt = len(self.sd.put_lists()[k][1])
        buf = np.zeros(cnt, dtype=np.float64)
        self._L.schwz_or_pack(self.h, k, _p(buf))
        return buf

    def unpack(self, k, buf):
        buf = np.ascontiguousarray(buf, dtype=np.float64)
        self._L.schwz_or_unpack(self.h, k, _p(buf))

    def update_boundary(self):
        self._L.schwz_or_update_boundary(self.h)

    def local_residual(self):
        return self._L.schwz_or_local_residual(self.h)

    def local_solve(self):
        return self._L.schwz_or_local_solve(self.h)

    def restrict(self):
        self._L.schwz_or_restrict(self.h)

    def set_local_max_iters(self, max_iters):
        self._L.schwz_or_state_set_local_max_iters(self.h, int(max_iters))

    def factors(self):
        ptrs = [C.c_void_p() for _ in range(7)]
        rc = self._L.schwz_or_state_factors(self.h, *[C.byref(p) for p in ptrs])
        if rc != 0:
            return None
        n = self.sd.local_size_x
        l_rp = _np_from(C.cast(ptrs[0], c_idx_p), n + 1, IDX)
        lnz = int(l_rp[-1])
        return dict(
            l_rp=l_rp, l_col=_np_from(C.cast(ptrs[1], c_idx_p), lnz, IDX),
            l_val=_np_from(C.cast(ptrs[2], c_dbl_p), lnz, np.float64),
            u_rp=_np_from(C.cast(ptrs[3], c_idx_p), n + 1, IDX),
            u_col=_np_from(C.cast(ptrs[4], c_idx_p), lnz, IDX),
            u_val=_np_from(C.cast(ptrs[5], c_dbl_p), lnz, np.float64),
            perm=_np_from(C.cast(ptrs[6], c_idx_p), n, IDX))

    def close(self):
        if self.h:
            self._L.schwz_or_state_free(self.h)
            self.h = None


# --------------------------------------------------------------------------
# whole run
# --------------------------------------------------------------------------

def ras_run(rp, col, val, rhs, P, first_row, settings, history=True):
    L = lib()
    N = len(rp) - 1
    sol = np.zeros(N, dtype=np.float64)
    mi = max(int(settings.max_iters), 1)
    hg = np.zeros(mi, dtype=np.float64) if history else None
    hl = np.zeros(mi * P, dtype=np.float64) if history else None
    hi = np.zeros(mi * P, dtype=np.int32) if history else None
    res = Result()
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    first_row = np.ascontiguousarray(first_row, dtype=IDX)
    rc = L.schwz_or_ras_run(N, _p(rp), _p(col), _p(val), _p(rhs), P, _p(first_row),
                            C.byref(settings), _p(sol), _p(hg), _p(hl), _p(hi),
                            C.byref(res))
    out = dict(rc=rc, solution=sol, iter_count=res.iter_count, converged=bool(res.converged),
               residual_norm=res.residual_norm, rhs_norm=res.rhs_norm, sol_norm=res.sol_norm,
               elapsed_s=res.elapsed_s, setup_s=res.setup_s)
    if history:
        k = min(res.iter_count + 1, mi)
        out["hist_global"] = hg[:k]
        out["hist_local"] = hl.reshape(mi, P)[:k]
        out["hist_inner"] = hi.reshape(mi, P)[:res.iter_count]
    return out


# --------------------------------------------------------------------------
# stand-alone kernels
# --------------------------------------------------------------------------

def spmv(rp, col, val, x, alpha=1.0, beta=0.0, y=None):
    n = len(rp) - 1
    if y is None:
        y = np.zeros(n, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().schwz_or_spmv(n, _p(rp), _p(col), _p(val), alpha, _p(x), beta, _p(y))
    return y


def gather(idx, src, into, op=OP_COPY):
    idx = np.ascontiguousarray(idx, dtype=IDX)
    lib().schwz_or_gather(len(idx), _p(idx), _p(src), _p(into), op)
    return into


def scatter(idx, src, into, op=OP_COPY):
    idx = np.ascontiguousarray(idx, dtype=IDX)
    lib().schwz_or_scatter(len(idx), _p(idx), _p(src), _p(into), op)
    return into


def jacobi_blocks(rp, col, max_block_size):
    """Block boundaries of the block-Jacobi preconditioner (Ginkgo's supervariable agglomeration)."""
    n = len(rp) - 1
    ptr = np.zeros(n + 2, dtype=np.int64)
    nb = lib().schwz_or_jacobi_blocks(n, _p(np.ascontiguousarray(rp, dtype=IDX)),
                                      _p(np.ascontiguousarray(col, dtype=IDX)), int(max_block_size), _p(ptr))
    return ptr[:nb + 1].copy()


def precond_code(local_precond, block_size=1):
    """(OR_PRECOND_*, block size) of a --local_precond / --precond_max_block_size pair
    (solve.cpp:488-571)."""
    if local_precond == "block-jacobi":
        return (1, 1) if int(block_size) == 1 else (2, int(block_size))
    if local_precond == "ilu":
        return 3, 1
    if local_precond == "isai":
        return 4, 1
    return 0, 1


def pcg(rp, col, val, b, x0=None, precond=PRECOND_NONE, rtol=1e-12, max_iters=-1, block_size=1):
    n = len(rp) - 1
    x = np.zeros(n, dtype=np.float64) if x0 is None else np.array(x0, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    rn = C.c_double(0.0)
    if max_iters < 0:
        max_iters = n
    it = lib().schwz_or_pcg_ex(n, _p(rp), _p(col), _p(val), _p(b), _p(x), precond, block_size, rtol,
                               max_iters, C.byref(rn))
    return x, it, rn.value


def gmres(rp, col, val, b, x0=None, precond=PRECOND_NONE, rtol=1e-12, max_iters=-1, restart=1, block_size=1):
    n = len(rp) - 1
    x = np.zeros(n, dtype=np.float64) if x0 is None else np.array(x0, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    rn = C.c_double(0.0)
    if max_iters < 0:
        max_iters = n
    it = lib().schwz_or_gmres(n, _p(rp), _p(col), _p(val), _p(b), _p(x), precond, block_size, restart, rtol,
                              max_iters, C.byref(rn))
    return x, it, rn.value


def ilu0(rp, col, val):
    L = lib()
    n = len(rp) - 1
    ptrs = [C.c_void_p() for _ in range(6)]
    L.schwz_or_ilu0(n, _p(rp), _p(col), _p(val), *[C.byref(p) for p in ptrs])
    l_rp = _np_from(C.cast(ptrs[0], c_idx_p), n + 1, IDX)
    u_rp = _np_from(C.cast(ptrs[3], c_idx_p), n + 1, IDX)
    out = dict(l_rp=l_rp, l_col=_np_from(C.cast(ptrs[1], c_idx_p), int(l_rp[-1]), IDX),
               l_val=_np_from(C.cast(ptrs[2], c_dbl_p), int(l_rp[-1]), np.float64),
               u_rp=u_rp, u_col=_np_from(C.cast(ptrs[4], c_idx_p), int(u_rp[-1]), IDX),
               u_val=_np_from(C.cast(ptrs[5], c_dbl_p), int(u_rp[-1]), np.float64))
    for p in ptrs:
        L.schwz_or_free(p)
    return out


def isai(rp, col, val, lower):
    L = lib()
    n = len(rp) - 1
    p = C.c_void_p()
    L.schwz_or_isai(n, _p(rp), _p(col), _p(val), int(bool(lower)), C.byref(p))
    out = _np_from(C.cast(p, c_dbl_p), int(rp[-1]), np.float64)
    L.schwz_or_free(p)
    return out


def cholesky(rp, col, val, natural=False):
    L = lib()
    n = len(rp) - 1
    ptrs = [C.c_void_p() for _ in range(7)]
    rc = L.schwz_or_cholesky(n, _p(rp), _p(col), _p(val), int(natural),
                             *[C.byref(p) for p in ptrs])
    l_rp = _np_from(C.cast(ptrs[0], c_idx_p), n + 1, IDX)
    lnz = int(l_rp[-1])
    out = dict(
        status=rc, l_rp=l_rp, l_col=_np_from(C.cast(ptrs[1], c_idx_p), lnz, IDX),
        l_val=_np_from(C.cast(ptrs[2], c_dbl_p), lnz, np.float64),
        u_rp=_np_from(C.cast(ptrs[3], c_idx_p), n + 1, IDX),
        u_col=_np_from(C.cast(ptrs[4], c_idx_p), lnz, IDX),
        u_val=_np_from(C.cast(ptrs[5], c_dbl_p), lnz, np.float64),
        perm=_np_from(C.cast(ptrs[6], c_idx_p), n, IDX))
    for p in ptrs:
        L.schwz_or_free(p)
    return out


def direct_solve(f, b):
    n = len(f["perm"])
    y = np.zeros(n, dtype=np.float64)
    w = np.zeros(2 * n, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    lib().schwz_or_direct_solve(n, _p(f["l_rp"]), _p(f["l_col"]), _p(f["l_val"]),
                                _p(f["u_rp"]), _p(f["u_col"]), _p(f["u_val"]), _p(f["perm"]),
                                _p(b), _p(y), _p(w))
    return y


def read_matrix_market(path):
    """Matrix-Market coordinate reader (gko::read + sort_by_column_index,
    initialization.cpp:204-213).  Pure text parsing; returns CSR int32/float64."""
    with open(path) as f:
        header = f.readline().lower().split()
        symmetric = "symmetric" in header
        pattern = "pattern" in header
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        nr, nc, nnz = [int(t) for t in line.split()]
        data = np.loadtxt(f, ndmin=2)
    r = data[:, 0].astype(np.int64) - 1
    c = data[:, 1].astype(np.int64) - 1
    v = np.ones(len(r)) if pattern else data[:, 2].astype(np.float64)
    if symmetric:
        off = r != c
        r, c, v = (np.concatenate([r, c[off]]), np.concatenate([c, r[off]]),
                   np.concatenate([v, v[off]]))
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    rp = np.zeros(nr + 1, dtype=IDX)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp).astype(IDX)
    return rp, c.astype(IDX), v.astype(np.float64)
