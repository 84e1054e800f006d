"""Thin object wrappers over the C ABI (include/schwz_hip.h).

Device buffers are plain integer addresses here (torch supplies them in the
host layer: `tensor.data_ptr()`); host arrays are numpy.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from ._capi import check, lib, ptr

IDX = np.int32


def _stream_arg(stream):
    return C.c_void_p(int(stream)) if stream else None


# ---------------------------------------------------------------------------
# stand-alone device objects
# ---------------------------------------------------------------------------

def gather(n, d_idx, d_from, d_into, op=capi.OP_COPY, stream=0):
    """gather_kernel.cu:46-109."""
    check(lib.schwz_gather(n, ptr(d_idx), ptr(d_from), ptr(d_into), op, _stream_arg(stream)))


def scatter(n, d_idx, d_from, d_into, op=capi.OP_COPY, stream=0):
    """scatter_kernel.cu:43-107."""
    check(lib.schwz_scatter(n, ptr(d_idx), ptr(d_from), ptr(d_into), op, _stream_arg(stream)))


class Csr:
    """A CSR matrix resident in HBM (gko::matrix::Csr on the device executor)."""

    def __init__(self, rp, col, val, ncols=None):
        rp = np.ascontiguousarray(rp, dtype=IDX)
        col = np.ascontiguousarray(col, dtype=IDX)
        val = np.ascontiguousarray(val, dtype=np.float64)
        self.nrows = len(rp) - 1
        self.ncols = self.nrows if ncols is None else ncols
        h = C.c_void_p()
        check(lib.schwz_csr_create(self.nrows, self.ncols, ptr(rp), ptr(col), ptr(val), C.byref(h)))
        self.h = h
        self.nnz = int(lib.schwz_csr_nnz(h))

    def spmv(self, d_x, d_y, alpha=1.0, beta=0.0, variant=0, stream=0):
        check(lib.schwz_csr_spmv(self.h, alpha, ptr(d_x), beta, ptr(d_y), variant,
                                 _stream_arg(stream)))

    def format(self):
        """Coding variant 0 uses: 0 plain CSR, 1 per-entry dictionaries, 2 row patterns, 3 row pairs."""
        return int(lib.schwz_csr_format(self.h))

    def symmetric(self):
        """True when the upload found the (row-pair coded) matrix symmetric bit for bit."""
        return bool(lib.schwz_csr_symmetric(self.h))

    def sweep_slots(self):
        """Workgroup slots of the z-sweep walk of the CG update launch (0: not a canonical 3-D stencil)."""
        return int(lib.schwz_csr_sweep_slots(self.h))

    def sweep_left_out(self):
        """Chunks of 512 rows the z-sweep walk leaves to its companion launch."""
        return int(lib.schwz_csr_sweep_left_out(self.h))

    def algorithmic_bytes(self):
        # 12 nnz + 4 (rows+1) + 16 rows
        return 12 * self.nnz + 4 * (self.nrows + 1) + 16 * self.nrows

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.schwz_csr_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Pcg:
    """Device-resident preconditioned CG (gko::solver::Cg + stop::Combined)."""

    def __init__(self, csr, precond=capi.PRECOND_NONE, block_size=1):
        self.csr = csr
        h = C.c_void_p()
        check(lib.schwz_pcg_create_ex(csr.h, precond, block_size, C.byref(h)))
        self.h = h

    def solve(self, d_b, d_x, rtol, max_iters, stream=0, want_stats=True):
        it = C.c_int(0)
        rn = C.c_double(0.0)
        check(lib.schwz_pcg_solve(self.h, ptr(d_b), ptr(d_x), rtol, max_iters,
                                  C.byref(it) if want_stats else None,
                                  C.byref(rn) if want_stats else None, _stream_arg(stream)))
        return it.value, rn.value

    def flavour(self):
        """How the last solve iterated (schwz_pcg_flavour)."""
        return int(lib.schwz_pcg_flavour(self.h))

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.schwz_pcg_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Gmres:
    """Device-resident restarted GMRES, right preconditioned (gko::solver::Gmres with
    krylov_dim = restart; solve.cpp:486-520)."""

    def __init__(self, csr, precond=capi.PRECOND_NONE, block_size=1, restart=1):
        self.csr = csr
        h = C.c_void_p()
        check(lib.schwz_gmres_create(csr.h, precond, block_size, restart, C.byref(h)))
        self.h = h

    def solve(self, d_b, d_x, rtol, max_iters, stream=0, want_stats=True):
        it = C.c_int(0)
        rn = C.c_double(0.0)
        check(lib.schwz_gmres_solve(self.h, ptr(d_b), ptr(d_x), rtol, max_iters,
                                    C.byref(it) if want_stats else None,
                                    C.byref(rn) if want_stats else None, _stream_arg(stream)))
        return it.value, rn.value

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.schwz_gmres_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Trs:
    """y = P^T L^-T L^-1 P b (gko LowerTrs/UpperTrs + Permutation)."""

    def __init__(self, l_rp, l_col, l_val, u_rp, u_col, u_val, perm=None):
        arrs = [np.ascontiguousarray(a, dtype=t) for a, t in
                ((l_rp, IDX), (l_col, IDX), (l_val, np.float64), (u_rp, IDX), (u_col, IDX),
                 (u_val, np.float64))]
        arrs.append(None if perm is None else np.ascontiguousarray(perm, dtype=IDX))
        self.n = len(l_rp) - 1
        h = C.c_void_p()
        check(lib.schwz_trs_create(self.n, *[ptr(a) for a in arrs], C.byref(h)))
        self.h = h

    def solve(self, d_b, d_y, stream=0):
        check(lib.schwz_trs_solve(self.h, ptr(d_b), ptr(d_y), _stream_arg(stream)))

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.schwz_trs_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def ilu0(rp, col, val):
    """Host ILU(0) standing in for gko::factorization::ParIlu (solve.cpp:506-532)."""
    rp = np.ascontiguousarray(rp, dtype=IDX)
    col = np.ascontiguousarray(col, dtype=IDX)
    val = np.ascontiguousarray(val, dtype=np.float64)
    n = len(rp) - 1
    out = [C.c_void_p() for _ in range(6)]
    check(lib.schwz_ilu0(n, ptr(rp), ptr(col), ptr(val), *[C.byref(o) for o in out]))

    def take(p, cnt, ctype, dtype):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=(max(cnt, 1),))[:cnt].copy().astype(dtype)

    l_rp = take(out[0], n + 1, C.c_int32, IDX)
    u_rp = take(out[3], n + 1, C.c_int32, IDX)
    res = dict(l_rp=l_rp, l_col=take(out[1], int(l_rp[-1]), C.c_int32, IDX),
               l_val=take(out[2], int(l_rp[-1]), C.c_double, np.float64), u_rp=u_rp,
               u_col=take(out[4], int(u_rp[-1]), C.c_int32, IDX),
               u_val=take(out[5], int(u_rp[-1]), C.c_double, np.float64))
    for o in out:
        lib.schwz_free(o)
    return res


def isai(rp, col, val, lower):
    """ISAI values of a triangular CSR factor on its own pattern (LowerIsai / UpperIsai,
    solve.cpp:616-638)."""
    rp = np.ascontiguousarray(rp, dtype=IDX)
    col = np.ascontiguousarray(col, dtype=IDX)
    val = np.ascontiguousarray(val, dtype=np.float64)
    n = len(rp) - 1
    out = C.c_void_p()
    check(lib.schwz_isai(n, ptr(rp), ptr(col), ptr(val), int(bool(lower)), C.byref(out)))
    nnz = int(rp[-1])
    w = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_double)), shape=(max(nnz, 1),))[:nnz].copy()
    lib.schwz_free(out)
    return w


def cholesky(rp, col, val, natural=False):
    """Host sparse LL^T standing in for CHOLMOD (solve.cpp:75-143)."""
    rp = np.ascontiguousarray(rp, dtype=IDX)
    col = np.ascontiguousarray(col, dtype=IDX)
    val = np.ascontiguousarray(val, dtype=np.float64)
    n = len(rp) - 1
    out = [C.c_void_p() for _ in range(7)]
    check(lib.schwz_cholesky(n, ptr(rp), ptr(col), ptr(val), int(natural),
                             *[C.byref(o) for o in out]))

    def take(p, cnt, ctype, dtype):
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=(max(cnt, 1),))[:cnt].copy()
        return a.astype(dtype, copy=False)

    l_rp = take(out[0], n + 1, C.c_int32, IDX)
    lnz = int(l_rp[-1]) if n else 0
    res = dict(l_rp=l_rp, l_col=take(out[1], lnz, C.c_int32, IDX),
               l_val=take(out[2], lnz, C.c_double, np.float64),
               u_rp=take(out[3], n + 1, C.c_int32, IDX),
               u_col=take(out[4], lnz, C.c_int32, IDX),
               u_val=take(out[5], lnz, C.c_double, np.float64),
               perm=take(out[6], n, C.c_int32, IDX))
    for o in out:
        lib.schwz_free(o)
    return res


class DeviceWindow:
    """A device buffer other rank processes of the node can map (MPI_Win_create of
    communicate.hpp:67-224 over HIP IPC).  `handle` travels to the peers over any host channel."""

    def __init__(self, count, single=False):
        self.count, self.esize = int(count), 4 if single else 8
        p = C.c_void_p()
        check(lib.schwz_window_alloc(self.count * self.esize, C.byref(p)))
        self.ptr = p.value
        buf = (C.c_ubyte * 64)()
        check(lib.schwz_window_export(C.c_void_p(self.ptr), buf))
        self.handle = ("hip-ipc", bytes(buf), self.count, self.esize)

    def at(self, offset):
        return self.ptr + int(offset) * self.esize

    def close(self):
        if getattr(self, "ptr", None) and lib is not None:
            lib.schwz_window_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        self.close()


class PeerWindow:
    """Another rank's DeviceWindow mapped into this process."""

    def __init__(self, handle):
        kind, raw, self.count, self.esize = handle
        assert kind == "hip-ipc"
        buf = (C.c_ubyte * 64).from_buffer_copy(raw)
        p = C.c_void_p()
        check(lib.schwz_window_open(buf, C.byref(p)))
        self.ptr = p.value

    def at(self, offset):
        return self.ptr + int(offset) * self.esize

    def close(self):
        if getattr(self, "ptr", None) and lib is not None:
            lib.schwz_window_close(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        self.close()


def host_atomic_add(arr, index, v=1):
    """MPI_Accumulate(MPI_SUM) on an int32 of a shared-memory window."""
    return int(lib.schwz_host_atomic_add_i32(C.c_void_p(arr.ctypes.data + 4 * int(index)), int(v)))


def host_atomic_min(arr, index, v):
    """MPI_Accumulate(MPI_MIN) on a float64 of a shared-memory window."""
    return float(lib.schwz_host_atomic_min_f64(C.c_void_p(arr.ctypes.data + 8 * int(index)), float(v)))


# ---------------------------------------------------------------------------
# host setup
# ---------------------------------------------------------------------------

class Problem:
    """Global system matrix as a row source (never replicated per rank)."""

    def __init__(self, handle):
        self.h = handle
        self.N = int(lib.schwz_problem_size(handle))
        self.nnz = int(lib.schwz_problem_nnz(handle))

    @classmethod
    def laplacian(cls, dim, nx, ny=None, nz=None):
        """initialization.cpp:214-265 (dim=2) and its 3-D 7-point extension."""
        ny = nx if ny is None else ny
        nz = (1 if dim == 2 else nx) if nz is None else nz
        h = C.c_void_p()
        check(lib.schwz_problem_laplacian(dim, nx, ny, nz, C.byref(h)))
        return cls(h)

    @classmethod
    def from_csr(cls, rp, col, val):
        rp = np.ascontiguousarray(rp, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=IDX)
        val = np.ascontiguousarray(val, dtype=np.float64)
        h = C.c_void_p()
        check(lib.schwz_problem_from_csr(len(rp) - 1, ptr(rp), ptr(col), ptr(val), C.byref(h)))
        return cls(h)

    @classmethod
    def from_rows(cls, N, row_ids, rp, col, val):
        """A row source that holds the rows `row_ids` (ascending global ids) of an N x N matrix only:
        what a rank receives in the distributed ingest (schwz_problem_from_rows)."""
        row_ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        rp = np.ascontiguousarray(rp, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=IDX)
        val = np.ascontiguousarray(val, dtype=np.float64)
        h = C.c_void_p()
        check(lib.schwz_problem_from_rows(int(N), len(row_ids), ptr(row_ids), ptr(rp), ptr(col), ptr(val),
                                          C.byref(h)))
        return cls(h)

    def extract_rows(self, row_ids):
        """(rp, col, val) of the rows `row_ids`, columns global (schwz_problem_extract_rows)."""
        row_ids = np.ascontiguousarray(row_ids, dtype=np.int64)
        rp = np.zeros(len(row_ids) + 1, dtype=np.int64)
        check(lib.schwz_problem_extract_rows(self.h, len(row_ids), ptr(row_ids), ptr(rp), None, None))
        col = np.zeros(max(int(rp[-1]), 1), dtype=IDX)
        val = np.zeros(max(int(rp[-1]), 1), dtype=np.float64)
        check(lib.schwz_problem_extract_rows(self.h, len(row_ids), ptr(row_ids), ptr(rp), ptr(col), ptr(val)))
        return rp, col[:rp[-1]], val[:rp[-1]]

    @classmethod
    def from_matrix_market(cls, path):
        """initialization.cpp:204-213."""
        h = C.c_void_p()
        check(lib.schwz_problem_from_matrix_market(path.encode(), C.byref(h)))
        return cls(h)

    def row(self, g):
        cap = 64
        while True:
            cols = np.zeros(cap, dtype=np.int64)
            vals = np.zeros(cap, dtype=np.float64)
            n = C.c_int(0)
            rc = lib.schwz_problem_row(self.h, g, C.byref(n), ptr(cols), ptr(vals), cap)
            if rc == capi.ERR_INVALID and cap < (1 << 24) and 0 <= g < self.N:
                cap *= 4
                continue
            check(rc)
            return cols[:n.value], vals[:n.value]

    def to_csr(self):
        """Materialise (small problems only; used by tests)."""
        rp = np.zeros(self.N + 1, dtype=np.int64)
        cols, vals = [], []
        for g in range(self.N):
            c, v = self.row(g)
            cols.append(c)
            vals.append(v)
            rp[g + 1] = rp[g] + len(c)
        return rp, np.concatenate(cols).astype(IDX), np.concatenate(vals)

    def permute(self, part, P):
        """restricted_schwarz.cpp:105-152; returns (problem, perm, first_row)."""
        part = np.ascontiguousarray(part, dtype=np.uint32)
        perm = np.zeros(self.N, dtype=np.int64)
        fr = np.zeros(P + 1, dtype=np.int64)
        h = C.c_void_p()
        check(lib.schwz_problem_permute(self.h, P, ptr(part), ptr(perm), ptr(fr), C.byref(h)))
        return Problem(h), perm, fr

    def partition_graph(self, P):
        part = np.zeros(self.N, dtype=np.uint32)
        check(lib.schwz_partition_graph(self.h, P, ptr(part)))
        return part

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.schwz_problem_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def rhs_random(global_ids):
    """Initialize::generate_rhs (initialization.cpp:88-96) by global row id."""
    ids = np.ascontiguousarray(global_ids, dtype=np.int64)
    out = np.zeros(max(len(ids), 1), dtype=np.float64)
    check(lib.schwz_rhs_random(len(ids), ptr(ids), ptr(out)))
    return out[:len(ids)]


def partition_regular(N, P):
    fr = np.zeros(P + 1, dtype=np.int64)
    check(lib.schwz_partition_regular(N, P, ptr(fr)))
    return fr


def partition_regular2d(n1d, P):
    part = np.zeros(n1d * n1d, dtype=np.uint32)
    check(lib.schwz_partition_regular2d(n1d, P, ptr(part)))
    return part


class Subdomain:
    """One subdomain: host index sets + (after to_device) the HBM-resident
    iteration state.  Mirrors what SolverRAS::setup_local_matrices /
    setup_comm_buffers build (restricted_schwarz.cpp:56-604)."""

    def __init__(self, problem, P, me, overlap, first_row):
        self.problem = problem
        self.P, self.me, self.overlap = P, me, overlap
        self.first_row = np.ascontiguousarray(first_row, dtype=np.int64)
        h = C.c_void_p()
        check(lib.schwz_subdomain_setup(problem.h, P, me, overlap, ptr(self.first_row),
                                        C.byref(h)))
        self.h = h
        self.on_device = False
        self._sizes()

    def _sizes(self):
        s = np.zeros(10, dtype=np.int64)
        check(lib.schwz_subdomain_sizes(self.h, ptr(s)))
        (self.local_size, self.local_size_x, self.overlap_size, self.halo_size, self.nnz_local,
         self.nnz_interface, self.num_neighbors_in, self.num_neighbors_out, self.num_recv,
         self.num_send) = [int(v) for v in s]

    @property
    def local_to_global(self):
        out = np.zeros(self.local_size_x + self.halo_size, dtype=np.int64)
        check(lib.schwz_subdomain_local_to_global(self.h, ptr(out)))
        return out

    def local_matrix(self):
        rp = np.zeros(self.local_size_x + 1, dtype=IDX)
        col = np.zeros(max(self.nnz_local, 1), dtype=IDX)
        val = np.zeros(max(self.nnz_local, 1), dtype=np.float64)
        check(lib.schwz_subdomain_local_matrix(self.h, ptr(rp), ptr(col), ptr(val)))
        return rp, col[:self.nnz_local], val[:self.nnz_local]

    def interface_matrix(self):
        rp = np.zeros(self.local_size_x + 1, dtype=IDX)
        col = np.zeros(max(self.nnz_interface, 1), dtype=np.int64)
        val = np.zeros(max(self.nnz_interface, 1), dtype=np.float64)
        check(lib.schwz_subdomain_interface_matrix(self.h, ptr(rp), ptr(col), ptr(val)))
        return rp, col[:self.nnz_interface], val[:self.nnz_interface]

    def _list(self, fn, k):
        rank = C.c_int(0)
        cnt = C.c_int64(0)
        check(fn(self.h, k, C.byref(rank), C.byref(cnt), None))
        ids = np.zeros(max(cnt.value, 1), dtype=np.int64)
        check(fn(self.h, k, None, None, ptr(ids)))
        return rank.value, ids[:cnt.value]

    def get_lists(self):
        return [self._list(lib.schwz_subdomain_get_list, k) for k in range(self.num_neighbors_in)]

    def put_lists(self):
        return [self._list(lib.schwz_subdomain_put_list, k) for k in range(self.num_neighbors_out)]

    def add_put_list(self, p, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        check(lib.schwz_subdomain_add_put_list(self.h, p, len(ids), ptr(ids)))
        self._sizes()

    def send_offsets(self):
        out = []
        for k in range(self.num_neighbors_out + 1):
            o = C.c_int64(0)
            check(lib.schwz_subdomain_send_offset(self.h, k, C.byref(o)))
            out.append(o.value)
        return out

    def recv_offsets(self):
        out = []
        for k in range(self.num_neighbors_in + 1):
            o = C.c_int64(0)
            check(lib.schwz_subdomain_recv_offset(self.h, k, C.byref(o)))
            out.append(o.value)
        return out

    def local_rhs(self, rhs_fn):
        """[rhs[interior]; rhs[overlap_row]] (initialization.cpp:349-355);
        rhs_fn maps an array of global ids to values."""
        return np.ascontiguousarray(rhs_fn(self.local_to_global[:self.local_size_x]),
                                    dtype=np.float64)

    # ---- device -------------------------------------------------------------

    def to_device(self, local_rhs, local_solver=capi.SOLVER_ITERATIVE,
                  precond=capi.PRECOND_NONE, local_tol=1e-12, local_max_iters=-1,
                  natural_factor_ordering=False, spmv_variant=0, precond_block_size=1,
                  non_symmetric=False, restart_iter=1):
        local_rhs = np.ascontiguousarray(local_rhs, dtype=np.float64)
        assert len(local_rhs) == self.local_size_x
        opt = capi.SolverOptions(local_solver, precond, local_tol, local_max_iters,
                                 int(natural_factor_ordering), spmv_variant, int(precond_block_size),
                                 int(bool(non_symmetric)), int(restart_iter))
        check(lib.schwz_subdomain_to_device(self.h, ptr(local_rhs), C.byref(opt)))
        self.on_device = True

    def pack(self, d_send, stream=0):
        check(lib.schwz_ras_pack(self.h, ptr(d_send), _stream_arg(stream)))

    def unpack(self, d_recv, stream=0):
        check(lib.schwz_ras_unpack(self.h, ptr(d_recv), _stream_arg(stream)))

    def pack_f32(self, d_send, stream=0):
        check(lib.schwz_ras_pack_f32(self.h, ptr(d_send), _stream_arg(stream)))

    def unpack_f32(self, d_recv, stream=0):
        check(lib.schwz_ras_unpack_f32(self.h, ptr(d_recv), _stream_arg(stream)))

    def early_pack_ok(self):
        """Whether the local solver records the event pack_early waits for (CG with put lists)."""
        return bool(lib.schwz_ras_early_pack_ok(self.h))

    def pack_early(self, d_send, single=False, stream=0):
        """The pack of the NEXT exchange beside the tail of the running solve (schwz_ras_pack_early):
        `stream` waits until the rows of the put lists are final, then gathers them from the solve's result."""
        check(lib.schwz_ras_pack_early(self.h, ptr(d_send), int(bool(single)), _stream_arg(stream)))

    def pack_neighbor(self, k, dst, single=False, stream=0):
        """Out-neighbour k's halo values to the device address `dst` (its receive window)."""
        check(lib.schwz_ras_pack_neighbor(self.h, int(k), ptr(dst), int(bool(single)), _stream_arg(stream)))

    def unpack_neighbor(self, k, src, single=False, stream=0):
        """In-neighbour k's halo values from the device address `src` (its send window)."""
        check(lib.schwz_ras_unpack_neighbor(self.h, int(k), ptr(src), int(bool(single)), _stream_arg(stream)))

    def update_boundary(self, stream=0):
        check(lib.schwz_ras_update_boundary(self.h, _stream_arg(stream)))

    def local_residual(self, stream=0):
        out = C.c_double(0.0)
        check(lib.schwz_ras_local_residual(self.h, C.byref(out), _stream_arg(stream)))
        return out.value

    def local_residual_launch(self, stream=0):
        check(lib.schwz_ras_local_residual_launch(self.h, _stream_arg(stream)))

    def check_and_solve_launch(self, stream=0):
        check(lib.schwz_ras_check_and_solve_launch(self.h, _stream_arg(stream)))

    def norm_sq_to_device(self, d_ptr, stream=0):
        """Square of the last check residual norm to the device double at `d_ptr`, on `stream`, as soon as that
        scalar is final (the solve enqueued behind the check is not waited for)."""
        check(lib.schwz_ras_norm_sq_to_device(self.h, C.c_void_p(int(d_ptr)), _stream_arg(stream)))

    def local_residual_wait(self):
        out = C.c_double(0.0)
        check(lib.schwz_ras_local_residual_wait(self.h, C.byref(out)))
        return out.value

    def last_inner_stats(self):
        """(inner iterations, final residual norm) of the last local solve -- settings.enable_logging
        (solve.cpp:751-771); synchronises the device."""
        it, rn = C.c_int(0), C.c_double(0.0)
        check(lib.schwz_ras_last_inner_stats(self.h, C.byref(it), C.byref(rn)))
        return it.value, rn.value

    def set_local_max_iters(self, max_iters):
        """Inner iteration cap of later local solves (two-stage criterion, solve.cpp:723-742)."""
        check(lib.schwz_ras_set_local_max_iters(self.h, int(max_iters)))

    def local_solve(self, stream=0, want_iters=False):
        it = C.c_int(0)
        check(lib.schwz_ras_local_solve(self.h, C.byref(it) if want_iters else None,
                                        _stream_arg(stream)))
        return it.value

    def restrict(self, stream=0):
        check(lib.schwz_ras_restrict(self.h, _stream_arg(stream)))

    def cg_flavour(self):
        """How the last local CG solve iterated (schwz_ras_cg_flavour: bits 0-1 launches per iteration,
        4 deferred x update, 8 / 16 z-sweep walk of the update / fused direction launch, 32 of the start)."""
        return int(lib.schwz_ras_cg_flavour(self.h))

    def vector(self, which):
        p = C.c_void_p()
        n = C.c_int64(0)
        check(lib.schwz_ras_vector(self.h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def get_interior(self, stream=0):
        out = np.zeros(max(self.local_size, 1), dtype=np.float64)
        check(lib.schwz_ras_get_interior(self.h, ptr(out), _stream_arg(stream)))
        return out[:self.local_size]

    def true_residual_sq(self, stream=0):
        out = C.c_double(0.0)
        check(lib.schwz_ras_true_residual_sq(self.h, C.byref(out), _stream_arg(stream)))
        return out.value

    def algorithmic_bytes(self, which=0):
        return int(lib.schwz_ras_algorithmic_bytes(self.h, which))

    def close(self):
        if getattr(self, "h", None) and lib is not None:
            lib.schwz_subdomain_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()
