"""TEST-ONLY backend for schwz_amd.SolverRAS: the per-subdomain steps are executed by the
CPU oracle on host memory.  It exists so that the multi-process host path (index handshake,
halo exchange order, global convergence rule, solution gather over torch.distributed) can be
exercised with the gloo backend on a machine without GPUs.  Never imported by the product."""
import ctypes

import numpy as np
import torch

import oracle as O


def _view(ptr, n):
    if n == 0:
        return np.zeros(0)
    return np.ctypeslib.as_array((ctypes.c_double * n).from_address(int(ptr)))


class OracleProblem:
    def __init__(self, rp, col, val):
        self.rp, self.col, self.val = rp, col, val
        self.N = len(rp) - 1
        self.nnz = int(rp[-1])

    def permute(self, part, P):
        perm, iperm, fr, rp, col, val = O.apply_partition(self.rp, self.col, self.val, part, P)
        return OracleProblem(rp, col, val), perm.astype(np.int64), fr.astype(np.int64)

    def to_csr(self):
        return self.rp, self.col, self.val


class ShmWindow:
    """TEST stand-in for core.DeviceWindow / PeerWindow: a float64 array on a /dev/shm file that
    several rank processes map (host memory plays the device's part in the CPU tests)."""

    _serial = 0

    def __init__(self, count=None, single=False, handle=None):
        import os
        assert not single, "the test backend exchanges fp64 halos only"
        if handle is None:
            ShmWindow._serial += 1
            self.path = "/dev/shm/schwz_test_%d_%d" % (os.getpid(), ShmWindow._serial)
            self.count = max(int(count), 1)
            self.arr = np.memmap(self.path, dtype=np.float64, mode="w+", shape=(self.count,))
            self.arr[:] = 0.0
            self.owner = True
        else:
            _, self.path, self.count = handle
            self.arr = np.memmap(self.path, dtype=np.float64, mode="r+", shape=(self.count,))
            self.owner = False
        self.handle = ("shm-file", self.path, self.count)

    def at(self, offset):
        return self.arr[int(offset):]

    def close(self):
        import os
        if getattr(self, "owner", False) and os.path.exists(self.path):
            os.unlink(self.path)
            self.owner = False

    def __del__(self):
        self.close()


class OracleSubdomain:
    def __init__(self, problem, P, me, overlap, first_row):
        self.problem = problem
        self.overlap = overlap
        self.sd = O.Subdomain(problem.rp, problem.col, problem.val, P, me, overlap,
                              np.asarray(first_row, dtype=np.int32))
        self.state = None
        self._refresh()

    def _refresh(self):
        for k in ("local_size", "local_size_x", "overlap_size", "halo_size", "nnz_local",
                  "nnz_interface", "num_neighbors_in", "num_neighbors_out", "num_recv", "num_send"):
            setattr(self, k, getattr(self.sd, k))

    @property
    def local_to_global(self):
        return self.sd.local_to_global.astype(np.int64)

    def get_lists(self):
        return [(r, ids.astype(np.int64)) for r, ids in self.sd.get_lists()]

    def put_lists(self):
        return [(r, ids.astype(np.int64)) for r, ids in self.sd.put_lists()]

    def add_put_list(self, p, ids):
        self.sd.add_put_list(p, np.asarray(ids, dtype=np.int32))
        self._refresh()

    def _offsets(self, lists):
        off = [0]
        for _, ids in lists:
            off.append(off[-1] + len(ids))
        return off

    def send_offsets(self):
        return self._offsets(self.sd.put_lists())

    def recv_offsets(self):
        return self._offsets(self.sd.get_lists())

    def local_rhs(self, rhs_fn):
        return np.ascontiguousarray(rhs_fn(self.local_to_global[:self.local_size_x]))

    def to_device(self, local_rhs, local_solver=0, precond=0, local_tol=1e-12, local_max_iters=-1,
                  natural_factor_ordering=False, spmv_variant=0, precond_block_size=1, non_symmetric=False,
                  restart_iter=1):
        # the oracle state extracts the local rhs from a global vector: rebuild the entries it
        # will read
        rhs = np.zeros(self.sd.N)
        rhs[self.sd.local_to_global[:self.local_size_x]] = np.asarray(local_rhs)
        self._rhs_global = rhs
        s = O.make_settings(overlap=self.overlap, local_solver=local_solver, precond=precond,
                            local_tol=local_tol, local_max_iters=local_max_iters,
                            natural_factor_ordering=int(natural_factor_ordering),
                            precond_block_size=int(precond_block_size), non_symmetric=int(bool(non_symmetric)),
                            restart_iter=int(restart_iter))
        self._settings = s
        self.state = O.State(self.sd, rhs, s)

    def pack(self, d_send, stream=0):
        buf = _view(d_send, self.num_send)
        off = self.send_offsets()
        for k in range(self.num_neighbors_out):
            buf[off[k]:off[k + 1]] = self.state.pack(k)

    def unpack(self, d_recv, stream=0):
        buf = _view(d_recv, self.num_recv)
        off = self.recv_offsets()
        for k in range(self.num_neighbors_in):
            self.state.unpack(k, buf[off[k]:off[k + 1]].copy())

    def pack_neighbor(self, k, dst, single=False, stream=0):
        v = self.state.pack(k)
        dst[:len(v)] = v

    def unpack_neighbor(self, k, src, single=False, stream=0):
        off = self.recv_offsets()
        n = off[k + 1] - off[k]
        self.state.unpack(k, np.array(src[:n], dtype=np.float64))

    def update_boundary(self, stream=0):
        self.state.update_boundary()

    def local_residual(self, stream=0):
        return self.state.local_residual()

    def set_local_max_iters(self, max_iters):
        self.state.set_local_max_iters(max_iters)

    def local_solve(self, stream=0, want_iters=False):
        return self.state.local_solve()

    def restrict(self, stream=0):
        self.state.restrict()

    def get_interior(self, stream=0):
        g = self.state.global_solution()
        lo = int(self.sd.local_to_global[0]) if self.local_size else 0
        return g[lo:lo + self.local_size].copy()

    def true_residual_sq(self, stream=0):
        rp, col, val = self.sd.local_matrix()
        x = self.state.global_solution()[self.sd.local_to_global[:self.local_size_x]]
        ax = O.spmv(rp, col, val, x)
        r = self._rhs_global[self.sd.local_to_global[:self.local_size]] - ax[:self.local_size]
        return float(np.dot(r, r))

    def algorithmic_bytes(self, which=0):
        return 0


class OracleBackend:
    name = "oracle-test"
    device = torch.device("cpu")

    def stream(self):
        return 0

    def empty(self, n, single=False):
        assert not single, "the test backend exchanges fp64 halos only"
        return torch.zeros(max(int(n), 1), dtype=torch.float64)

    def synchronize(self):
        pass

    @staticmethod
    def window(count, single=False):
        return ShmWindow(count, single)

    @staticmethod
    def open_window(handle):
        return ShmWindow(handle=handle)

    @staticmethod
    def atomic_add(arr, index, v=1):
        from schwz_amd import core
        return core.host_atomic_add(arr, index, v)

    @staticmethod
    def atomic_min(arr, index, v):
        from schwz_amd import core
        return core.host_atomic_min(arr, index, v)

    @staticmethod
    def problem_laplacian(dim, nx, ny=None, nz=None):
        if dim == 2:
            return OracleProblem(*O.laplacian2d(nx))
        return OracleProblem(*O.laplacian3d(nx, ny, nz))

    @staticmethod
    def problem_from_csr(rp, col, val):
        return OracleProblem(np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(col, dtype=np.int32),
                             np.ascontiguousarray(val, dtype=np.float64))

    @staticmethod
    def problem_from_matrix_market(path):
        return OracleProblem(*O.read_matrix_market(path))

    @staticmethod
    def partition_regular(N, P):
        return O.first_rows_regular(N, P).astype(np.int64)

    @staticmethod
    def partition_regular2d(n1d, P):
        return O.partition_regular2d(n1d, P)

    def subdomain(self, problem, P, me, overlap, first_row):
        return OracleSubdomain(problem, P, me, overlap, first_row)

