"""Host-side mirror of the reference solver API for the RAS hot path.

`Settings`, `Metadata` and `SolverRAS(settings, metadata).initialize() / .run()`
keep the names, argument meaning and error behaviour of
include/settings.hpp:77-496, include/restricted_schwarz.hpp:61-107 and
SchwarzBase::initialize / run (source/schwarz_base.cpp:128-271, 323-506), as
driven by benchmarking/bench_ras.cpp:48-197.  All arithmetic runs in
libschwz_hip.so on the GPU; this file only orders the launches and talks to
the communication layer.
"""
import os
import time
from dataclasses import dataclass, field

import numpy as np

from . import _capi as capi
from . import core
from .comm import InProcessComm

# Settings::partition_settings / local_solver_settings (include/settings.hpp:96-156)
PARTITION_REGULAR = "regular"
PARTITION_REGULAR2D = "regular2d"
PARTITION_METIS = "metis"
PARTITION_CUSTOM = "custom"
SOLVER_DIRECT_CHOLMOD = "direct-cholmod"
SOLVER_DIRECT_UMFPACK = "direct-umfpack"
SOLVER_DIRECT_GINKGO = "direct-ginkgo"
SOLVER_ITERATIVE_GINKGO = "iterative-ginkgo"

# MEASURE_ELAPSED_FUNC_TIME ids and names (schwarz_base.cpp:393-450)
NEVER = (1 << 62)  # "no stop iteration agreed yet" in the flooded convergence state

TIMING_NAMES = ["boundary_exchange", "boundary_update", "convergence_check", "local_solve",
                "expand_local_vec"]


@dataclass
class CommSettings:
    enable_onesided: bool = False
    enable_overlap: bool = False
    enable_put: bool = False
    enable_get: bool = True
    stage_through_host: bool = False
    enable_one_by_one: bool = False
    enable_flush_local: bool = False
    enable_flush_all: bool = True
    enable_lock_local: bool = False
    enable_lock_all: bool = True


@dataclass
class ConvergenceSettings:
    put_all_local_residual_norms: bool = True
    enable_global_simple_tree: bool = False
    enable_decentralized_leader_election: bool = False
    enable_global_check: bool = True
    enable_accumulate: bool = False
    enable_global_check_iter_offset: bool = False
    convergence_crit: str = "solution_based"


@dataclass
class Settings:
    executor_string: str = "hip"
    matrix_filename: str = "null"
    explicit_laplacian: bool = True
    use_mixed_precision: bool = False
    enable_random_rhs: bool = False
    print_matrices: bool = False
    debug_print: bool = False
    partition: str = PARTITION_REGULAR
    local_solver: str = SOLVER_ITERATIVE_GINKGO
    non_symmetric_matrix: bool = False
    restart_iter: int = 1
    reset_local_crit_iter: int = -1
    overlap: int = 2
    naturally_ordered_factor: bool = False
    metis_objtype: str = ""
    write_debug_out: bool = False
    write_iters_and_residuals: bool = False
    enable_logging: bool = False
    write_perm_data: bool = False
    shifted_iter: int = 1
    factorization: str = "cholmod"
    reorder: str = ""
    comm_settings: CommSettings = field(default_factory=CommSettings)
    convergence_settings: ConvergenceSettings = field(default_factory=ConvergenceSettings)
    # extension (the reference has a 2-D generator only): 2 or 3, and
    # an optional (nx, ny, nz) for non-cubic 3-D grids
    laplacian_dim: int = 2
    laplacian_shape: tuple = None
    # optional user partition vector for PARTITION_CUSTOM
    partition_vector: object = None
    spmv_variant: int = 0


@dataclass
class Metadata:
    global_size: int = 0
    oned_laplacian_size: int = 0
    local_size: int = 0
    local_size_x: int = 0
    local_size_o: int = 0
    overlap_size: int = 0
    num_subdomains: int = 1
    my_rank: int = 0
    my_local_rank: int = 0
    local_num_procs: int = 1
    comm_size: int = 1
    num_threads: int = 1
    iter_count: int = 0
    tolerance: float = 1e-6
    local_solver_tolerance: float = 1e-12
    max_iters: int = 100
    local_max_iters: int = -1
    updated_max_iters: int = -1
    local_precond: str = "null"
    precond_max_block_size: int = 16
    current_residual_norm: float = -1.0
    min_residual_norm: float = -1.0
    time_struct: list = field(default_factory=list)
    comm_data_struct: list = field(default_factory=list)
    post_process_data: dict = field(default_factory=lambda: dict(
        global_residual_vector_out=[], local_residual_vector_out=[],
        local_converged_iter_count=[], local_converged_resnorm=[], local_timestamp=[]))
    first_row: object = None
    permutation: object = None


class HipBackend:
    """Device memory, streams and subdomain objects for `--executor=hip`."""

    name = "hip"

    def __init__(self, device_index=0):
        import torch
        if capi.device_count() < 1 or not torch.cuda.is_available():
            raise capi.SchwzError(capi.ERR_HIP,
                                  "executor 'hip' needs a GPU: no HIP device is visible and "
                                  "there is no CPU fallback")
        self._torch = torch
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        capi.check(capi.lib.schwz_set_device(device_index))

    def stream(self):
        return self._torch.cuda.current_stream().cuda_stream

    def empty(self, n, single=False):
        return self._torch.empty(max(int(n), 1), device=self.device,
                                 dtype=self._torch.float32 if single else self._torch.float64)

    def synchronize(self):
        self._torch.cuda.synchronize()

    # halo windows of the free-running one-sided mode (HIP IPC)
    window = staticmethod(core.DeviceWindow)
    open_window = staticmethod(core.PeerWindow)
    atomic_add = staticmethod(core.host_atomic_add)
    atomic_min = staticmethod(core.host_atomic_min)

    # problem sources / partitions (host code of libschwz_hip.so)
    problem_laplacian = staticmethod(core.Problem.laplacian)
    problem_from_matrix_market = staticmethod(core.Problem.from_matrix_market)
    problem_from_csr = staticmethod(core.Problem.from_csr)
    problem_from_rows = staticmethod(core.Problem.from_rows)
    partition_regular = staticmethod(core.partition_regular)
    partition_regular2d = staticmethod(core.partition_regular2d)

    def subdomain(self, problem, P, me, overlap, first_row):
        return core.Subdomain(problem, P, me, overlap, first_row)


def _ratio(a, b):
    """a / b as the reference's double arithmetic gives it (solve.cpp:906-915): a zero initial
    residual yields NaN (0/0) or inf, never an exception, and NaN compares false against the
    tolerance, so such a run simply keeps iterating like the reference and the C++ mirror."""
    if b == 0.0:
        return float("nan") if a == 0.0 or a != a else float("inf")
    return a / b


def _local_solver_code(settings):
    ls = settings.local_solver
    if ls == SOLVER_ITERATIVE_GINKGO:
        return capi.SOLVER_ITERATIVE
    if ls in (SOLVER_DIRECT_GINKGO, SOLVER_DIRECT_CHOLMOD):
        # direct-cholmod runs CHOLMOD's host solve in the reference
        # (solve.cpp:688-694); the same P^T L^-T L^-1 P is applied on the GPU here
        return capi.SOLVER_DIRECT
    raise capi.NotImplementedSchwz(capi.ERR_NOT_IMPLEMENTED,
                                   "local solver '%s' is not implemented" % ls)


def _precond_code(metadata):
    lp = metadata.local_precond
    if lp in ("null", "", None):
        return capi.PRECOND_NONE
    if lp == "block-jacobi":
        bs = int(metadata.precond_max_block_size)
        if bs == 1:
            return capi.PRECOND_JACOBI
        if 1 < bs <= 32:
            return capi.PRECOND_BLOCK_JACOBI
        raise capi.NotImplementedSchwz(capi.ERR_NOT_IMPLEMENTED,
                                       "block-jacobi: precond_max_block_size must be in 1..32")
    if lp == "ilu":
        return capi.PRECOND_ILU
    if lp == "isai":
        return capi.PRECOND_ISAI
    # unknown names only print to stderr in the reference (solve.cpp:568-570); refuse instead
    raise capi.NotImplementedSchwz(
        capi.ERR_NOT_IMPLEMENTED,
        "local_precond '%s' is not implemented; available: null, block-jacobi, ilu, isai" % lp)


class SolverRAS:
    """Restricted Additive Schwarz solver (include/restricted_schwarz.hpp:61-107).

    One process drives the subdomains listed in `comm.local_ranks`: all of them
    (InProcessComm, one GPU) or exactly one (TorchDistComm, one GPU per rank).
    """

    def __init__(self, settings, metadata, comm=None, backend=None, quiet=False):
        self.settings = settings
        self.metadata = metadata
        self.quiet = quiet
        if settings.executor_string not in ("hip", "cuda"):
            # the reference's "reference"/"omp" executors are CPU paths; this
            # build has none (the CPU restatement lives in oracle/ for tests only)
            raise capi.NotImplementedSchwz(
                capi.ERR_NOT_IMPLEMENTED,
                "executor '%s' is not available: only 'hip' (MI355X) exists and there is "
                "no CPU fallback" % settings.executor_string)
        self._user_matrix, self._user_rhs = None, None
        self.comm = comm if comm is not None else InProcessComm(max(metadata.num_subdomains, 1))
        self.backend = backend if backend is not None else HipBackend(
            getattr(self.comm, "device_index", 0))
        metadata.num_subdomains = self.comm.size
        metadata.comm_size = self.comm.size
        metadata.my_rank = self.comm.rank
        self.subdomains = {}
        self.speculative_solve = True
        self.problem = None
        self.result = None

    def _print(self, *a):
        if self.comm.is_root and not self.quiet:
            print(*a, flush=True)

    # ------------------------------------------------------------------ setup
    def _setup_global_matrix(self):
        """Initialize::setup_global_matrix (initialization.cpp:197-272)."""
        s, m, be = self.settings, self.metadata, self.backend
        if self._user_matrix is not None:
            prob = be.problem_from_csr(*self._user_matrix)
            self._print("Matrix handed over by the caller ")
        elif s.matrix_filename != "null":
            prob = be.problem_from_matrix_market(s.matrix_filename)
            self._print("Matrix from file " + s.matrix_filename)
        elif s.explicit_laplacian:
            n = int(m.oned_laplacian_size)
            if s.laplacian_dim == 3:
                nx, ny, nz = s.laplacian_shape if s.laplacian_shape else (n, n, n)
                prob = be.problem_laplacian(3, nx, ny, nz)
                self._print("Laplacian 3D Matrix %dx%dx%d (generated in house) " % (nx, ny, nz))
            else:
                prob = be.problem_laplacian(2, n)
                self._print("Laplacian 2D Matrix (generated in house) ")
        else:
            raise capi.SchwzError(capi.ERR_IO, " Need to provide a matrix or enable the default "
                                               "laplacian matrix.")
        m.global_size = prob.N
        return prob

    def _distributed_ingest(self):
        """A matrix FILE on several rank processes: the root parses and partitions, every other rank
        receives the rows its subdomain reads (SCHWZ_DISTRIBUTED_INGEST=0: every rank parses the file and
        keeps the whole matrix, like the reference, initialization.cpp:204-213)."""
        s, comm = self.settings, self.comm
        return (self._user_matrix is None and s.matrix_filename != "null" and comm.size > 1
                and len(comm.local_ranks) == 1 and hasattr(comm, "scatter_objects")
                and hasattr(self.backend, "problem_from_rows")
                and os.environ.get("SCHWZ_DISTRIBUTED_INGEST", "1") != "0")

    def _ingest_distributed(self):
        """Root: Matrix-Market file -> partition / permutation -> for every rank the interior and overlap
        rows of its subdomain (what schwz_subdomain_setup reads), cut out of the permuted matrix.  Every
        rank: a row source over its own part only (schwz_problem_from_rows); sizes, first_row and the
        permutation are broadcast.  Nothing of global length but the permutation lives on a non-root rank."""
        s, m, be, comm = self.settings, self.metadata, self.backend, self.comm
        P = m.num_subdomains
        meta, pieces = None, None
        if comm.is_root:
            full = be.problem_from_matrix_market(s.matrix_filename)
            self._print("Matrix from file " + s.matrix_filename + " (parsed on the root, rows distributed)")
            m.global_size = full.N
            full = self._partition(full)
            pieces = []
            for r in range(P):
                host_sd = core.Subdomain(full, P, r, s.overlap, m.first_row)
                rows = np.sort(np.asarray(host_sd.local_to_global[:host_sd.local_size_x], dtype=np.int64))
                pieces.append((rows,) + tuple(full.extract_rows(rows)))
                host_sd.close()
            meta = (full.N, np.asarray(m.first_row, dtype=np.int64),
                    None if m.permutation is None else np.asarray(m.permutation, dtype=np.int64))
            full.close()
        N, first_row, perm = comm.broadcast_object(meta)
        rows, rp, col, val = comm.scatter_objects(pieces)
        m.global_size, m.first_row, m.permutation = int(N), first_row, perm
        return be.problem_from_rows(int(N), rows, rp, col, val)

    def _partition(self, prob):
        """Initialize::partition (initialization.cpp:278-329) + the first_row /
        permutation part of setup_local_matrices (restricted_schwarz.cpp:84-152)."""
        s, m, be = self.settings, self.metadata, self.backend
        P = m.num_subdomains
        first_row = be.partition_regular(prob.N, P)
        perm = None
        if s.partition == PARTITION_REGULAR:
            self._print(" Regular 1D partition")
        elif s.partition in (PARTITION_REGULAR2D, PARTITION_METIS, PARTITION_CUSTOM):
            if s.partition == PARTITION_REGULAR2D:
                self._print(" Regular 2D partition")
                part = be.partition_regular2d(int(round(prob.N ** 0.5)), P) if P > 1 else None
            elif s.partition == PARTITION_METIS:
                self._print(" METIS partition")
                part = prob.partition_graph(P) if P > 1 else None
            else:
                part = s.partition_vector
                if P > 1:
                    if part is None:
                        raise capi.SchwzError(capi.ERR_INVALID,
                                              "partition 'custom' needs settings.partition_vector")
                    part = np.asarray(part)
                    if part.shape != (prob.N,) or part.min() < 0 or part.max() >= P:
                        raise capi.SchwzError(capi.ERR_INVALID,
                                              "partition_vector must hold one part id in [0, %d) per row "
                                              "(%d rows)" % (P, prob.N))
            if P > 1 and s.write_debug_out and self.comm.is_root and part is not None:
                # partition_tools.hpp:96-106
                with open("part_indices.csv", "w") as f:
                    f.write("idx,subd\n")
                    f.writelines("%d,%d\n" % (i, p) for i, p in enumerate(np.asarray(part)))
            if P > 1:
                prob, perm, first_row = prob.permute(part, P)
        else:
            raise capi.NotImplementedSchwz(capi.ERR_NOT_IMPLEMENTED,
                                           "partition '%s' is not implemented" % s.partition)
        m.first_row = first_row
        m.permutation = perm
        return prob

    def _rhs(self, ids):
        """rhs = 1.0, or the default-seeded uniform(0,1) sequence when enable_random_rhs is set
        together with explicit_laplacian (schwarz_base.cpp:169-173, initialization.cpp:88-96).
        As in the reference the rhs is indexed by the (possibly permuted) row id.  A right-hand
        side handed to initialize() is indexed by the caller's (old) row id."""
        if self._user_rhs is not None:
            perm = self.metadata.permutation
            ids = np.asarray(ids, dtype=np.int64)
            return self._user_rhs[ids if perm is None else np.asarray(perm, dtype=np.int64)[ids]]
        if self.settings.enable_random_rhs and self.settings.explicit_laplacian \
                and self.settings.matrix_filename == "null":
            return core.rhs_random(ids)
        return np.ones(len(ids), dtype=np.float64)

    def _rhs_is_ones(self):
        return self._user_rhs is None and not (self.settings.enable_random_rhs and self.settings.explicit_laplacian
                                                and self.settings.matrix_filename == "null")

    def _rhs_first_rows(self, sd, count):
        """rhs of the first `count` local rows of a subdomain; the all-ones default needs no global ids (the
        local-to-global list of a 16.8 M-row subdomain is 134 MB to copy out of the library)."""
        if self._rhs_is_ones():
            return np.ones(int(count), dtype=np.float64)
        return np.ascontiguousarray(self._rhs(sd.local_to_global[:count]), dtype=np.float64)

    def initialize(self, matrix=None, rhs=None):
        """SchwarzBase::initialize (schwarz_base.cpp:128-271).  `matrix` = (row_ptr, col, val) of
        the global system and `rhs` (ones when None), the same on every rank: the analogue of
        the reference's initialize(dealii::SparseMatrix, dealii::Vector) overload
        (include/schwarz_base.hpp:96-97) for callers that assemble their own system."""
        s, m, be, comm = self.settings, self.metadata, self.backend, self.comm
        self._user_matrix = matrix
        self._user_rhs = None if rhs is None else np.ascontiguousarray(rhs, dtype=np.float64)
        solver_code = _local_solver_code(s)
        precond_code = _precond_code(m)
        if s.non_symmetric_matrix and solver_code != capi.SOLVER_ITERATIVE:
            raise capi.NotImplementedSchwz(capi.ERR_NOT_IMPLEMENTED,
                                           "non_symmetric_matrix needs the iterative local solver (GMRES); "
                                           "the direct path is an LL^T factorization")
        if self._distributed_ingest():
            prob = self._ingest_distributed()
        else:
            prob = self._setup_global_matrix()
            prob = self._partition(prob)
        self.problem = prob
        P = m.num_subdomains
        for me in comm.local_ranks:
            self.subdomains[me] = be.subdomain(prob, P, me, s.overlap, m.first_row)
        puts = comm.handshake({me: sd.get_lists() for me, sd in self.subdomains.items()})
        for me, sd in self.subdomains.items():
            for q, ids in puts[me]:
                sd.add_put_list(q, ids)
        self.send_buf, self.recv_buf = {}, {}
        for me, sd in self.subdomains.items():
            sd.to_device(self._rhs_first_rows(sd, sd.local_size_x), local_solver=solver_code, precond=precond_code,
                         local_tol=m.local_solver_tolerance, local_max_iters=m.local_max_iters,
                         natural_factor_ordering=s.naturally_ordered_factor,
                         spmv_variant=s.spmv_variant, precond_block_size=m.precond_max_block_size,
                         non_symmetric=s.non_symmetric_matrix, restart_iter=s.restart_iter)
            # use_mixed_precision (MixedValueType = float): halos travel as fp32
            self.send_buf[me] = be.empty(sd.num_send, s.use_mixed_precision)
            self.recv_buf[me] = be.empty(sd.num_recv, s.use_mixed_precision)
        if s.print_matrices or s.write_perm_data or s.debug_print:
            self._debug_dumps(solver_code)
        first = self.subdomains[comm.local_ranks[0]]
        m.local_size, m.local_size_x = first.local_size, first.local_size_x
        m.overlap_size, m.local_size_o = first.overlap_size, m.global_size
        # per-pair views into the packed buffers, neighbour order
        self._sends, self._recvs = {}, {}
        for me, sd in self.subdomains.items():
            off = sd.send_offsets()
            for k, (q, _) in enumerate(sd.put_lists()):
                self._sends[(me, q)] = self.send_buf[me][off[k]:off[k + 1]]
            off = sd.recv_offsets()
            for k, (p, _) in enumerate(sd.get_lists()):
                self._recvs[(p, me)] = self.recv_buf[me][off[k]:off[k + 1]]
        # out- / in-neighbour ranks per subdomain (the overlapped mode's flag messages go along these edges;
        # put_lists() / get_lists() copy the id lists out of the library: not per iteration)
        self._neighbour_ranks = {me: ([q for q, _ in sd.put_lists()], [p for p, _ in sd.get_lists()])
                                 for me, sd in self.subdomains.items()}
        m.comm_data_struct = [
            (me, [(q, len(ids)) for q, ids in sd.put_lists()],
             [(p, len(ids)) for p, ids in sd.get_lists()], sd.num_send, sd.num_recv)
            for me, sd in self.subdomains.items()]
        self.close()  # windows of an earlier initialize()
        self._win = None
        if s.comm_settings.enable_onesided and getattr(comm, "node_windows", False) \
                and not s.comm_settings.enable_overlap:
            self._setup_windows()
        self._print(" Problem size: %d, subdomains: %d, local size (rank %d): %d (+%d overlap)" %
                    (m.global_size, P, comm.rank, m.local_size, m.overlap_size))
        if solver_code == capi.SOLVER_ITERATIVE:
            lmi = m.local_size_x if m.local_max_iters == -1 else m.local_max_iters
            self._print(" Local max iters %d with restart iter %d" % (lmi, s.restart_iter))
        else:
            self._print(" Local direct solve with HIP TRS")

    def _debug_dumps(self, solver_code):
        """The reference's debug output for executors other than "cuda" (schwarz_base.cpp:252-257, solve.cpp:401-450,
        utils.cpp:94-108): print_matrices -> local_mat_<rank>.csv / int_mat_<rank>.csv (and L_mat / U_mat of the
        direct local solver) as 1-based "row,col,value" lines; write_perm_data -> perm_<rank>.csv / inv_perm_<rank>.csv
        of the factor ordering; debug_print -> the permutation check."""
        s = self.settings

        def dump(name, me, rp, col, val):
            rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
            with open("%s_%d.csv" % (name, me), "w") as f:
                f.writelines("%d,%d,%.17g\n" % (r + 1, c + 1, v) for r, c, v in zip(rows, col, val))

        for me, sd in self.subdomains.items():
            rp, col, val = sd.local_matrix()
            if s.print_matrices:
                dump("local_mat", me, rp, col, val)
                dump("int_mat", me, *sd.interface_matrix())
            if solver_code != capi.SOLVER_DIRECT:
                continue
            f = core.cholesky(rp, col, val, s.naturally_ordered_factor)
            if s.print_matrices:
                dump("U_mat", me, f["u_rp"], f["u_col"], f["u_val"])
                dump("L_mat", me, f["l_rp"], f["l_col"], f["l_val"])
            perm = np.asarray(f["perm"])
            inv = np.empty_like(perm)
            inv[perm] = np.arange(len(perm), dtype=perm.dtype)
            if s.debug_print:
                ok = np.array_equal(np.sort(perm), np.arange(len(perm)))
                self._print(" Rank %d Permutation is %s\n Rank %d Inverse Permutation is %s" %
                            (me, "correct" if ok else "incorrect", me, "correct" if ok else "incorrect"))
            if s.write_perm_data:
                np.savetxt("perm_%d.csv" % me, perm, fmt="%d")
                np.savetxt("inv_perm_%d.csv" % me, inv, fmt="%d")

    # -------------------------------------------------------------------- run
    def _log_local_solve(self, locals_):
        """settings.enable_logging (solve.cpp:751-771): inner iteration count, final inner residual
        norm and a timestamp per outer iteration.  Reads the solver state, i.e. synchronises."""
        if not self.settings.enable_logging:
            return
        ppd = self.metadata.post_process_data
        for _, sd in locals_:
            if hasattr(sd, "last_inner_stats"):
                it, rn = sd.last_inner_stats()
                ppd["local_converged_iter_count"].append(it)
                ppd["local_converged_resnorm"].append(rn)
                ppd["local_timestamp"].append(time.perf_counter() - self._t_begin)

    def _two_stage(self):
        """solve.cpp:723-742: once iter_count > reset_local_crit_iter the local stopping
        criterion is rebuilt with updated_max_iters (-1: local_size_x) as its iteration cap."""
        s, m = self.settings, self.metadata
        if (s.reset_local_crit_iter != -1 and m.iter_count > s.reset_local_crit_iter
                and not self._two_stage_on):
            for _, sd in self.subdomains.items():
                sd.set_local_max_iters(m.updated_max_iters)
            self._two_stage_on = True

    def _pack(self, me, sd, stream):
        if self.settings.use_mixed_precision:
            sd.pack_f32(self.send_buf[me].data_ptr(), stream)
        else:
            sd.pack(self.send_buf[me].data_ptr(), stream)

    def _unpack(self, me, sd, stream):
        if self.settings.use_mixed_precision:
            sd.unpack_f32(self.recv_buf[me].data_ptr(), stream)
        else:
            sd.unpack(self.recv_buf[me].data_ptr(), stream)

    def _exchange(self):
        stream = self.backend.stream()
        if getattr(self, "_early", None) is not None:
            # this exchange was posted beside the tail of the last local solves (_post_early_exchange):
            # the same values, already on their way
            handle, self._early = self._early, None
            self.comm.finish_exchange(handle)
            for me, sd in self.subdomains.items():
                self._unpack(me, sd, stream)
            return
        for me, sd in self.subdomains.items():
            self._pack(me, sd, stream)
        self.comm.exchange(self._sends, self._recvs)
        for me, sd in self.subdomains.items():
            self._unpack(me, sd, stream)

    def _early_exchange_ok(self):
        """The synchronous two-sided loop may post the exchange of iteration k + 1 beside the tail of the
        local solves of iteration k: the solvers make the rows of the put lists final first and record an
        event (schwz_ras_pack_early), the communicator sends device buffers from a side stream.  The values
        are those the exchange at the start of iteration k + 1 would read after the restriction: the
        iteration is the same bit for bit.  SCHWZ_EARLY_EXCHANGE=0 switches it off."""
        if getattr(self, "_early_ok", None) is None:
            comm = self.comm
            self._early_ok = (os.environ.get("SCHWZ_EARLY_EXCHANGE", "1") != "0"
                              and self.metadata.num_subdomains > 1
                              and getattr(comm, "supports_early_exchange", lambda: False)()
                              and all(hasattr(sd, "early_pack_ok") and sd.early_pack_ok()
                                      for sd in self.subdomains.values()))
        return self._early_ok

    def _post_early_exchange(self):
        single = self.settings.use_mixed_precision

        def pack(stream):
            for me, sd in self.subdomains.items():
                sd.pack_early(self.send_buf[me].data_ptr(), single, stream)
        self._early = self.comm.start_exchange_early(self._sends, self._recvs, pack)

    def begin_run(self):
        """State of SchwarzBase::run before its loop (schwarz_base.cpp:340-386)."""
        m, P = self.metadata, self.metadata.num_subdomains
        cs, cv = self.settings.comm_settings, self.settings.convergence_settings
        if cs.enable_onesided and not (cv.enable_global_simple_tree or
                                       cv.enable_decentralized_leader_election):
            raise capi.SchwzError(capi.ERR_INVALID, "Global Convergence check type unspecified")
        if cs.enable_onesided and cs.enable_overlap and P > 62:
            # the flooded state carries one bit per subdomain in an int64 (the mirror refuses the
            # same, host/src/schwarz_base.cpp)
            raise capi.NotImplementedSchwz(capi.ERR_NOT_IMPLEMENTED,
                                           "overlapped one-sided mode supports at most 62 subdomains")
        self._announce_protocol()
        self._lres0 = {me: -1.0 for me in self.subdomains}
        self._flags = {me: False for me in self.subdomains}
        self._gres, self._gres0 = 0.0, -1.0
        self._num_converged = 0
        self._timings = [[] for _ in range(5)]
        # overlapped one-sided mode: flooded convergence state and the messages in flight
        self._mask = {me: 0 for me in self.subdomains}
        self._stop = {me: NEVER for me in self.subdomains}
        self._pending = None
        if getattr(self, "_early", None) is not None:  # an exchange left over from an interrupted run
            self._exchange()
        self._early = None
        self._t_begin = time.perf_counter()
        if getattr(self, "_win", None) is not None:
            # a fresh run: the windows go back to their initial state (collectively, before anybody iterates)
            self.comm.barrier()
            hw = self._win["host"]
            if self.comm.is_root:
                hw["tree"][:] = 0
                hw["flags"][:] = 0
                hw["count"][:] = 0
                hw["resid"][:] = np.finfo(np.float64).max
            self._win["sent"][:] = 0
            self._win["counted"] = False
            self.comm.barrier()
        if getattr(self, "_two_stage_on", False):  # a re-run starts with the first-stage cap again
            for _, sd in self.subdomains.items():
                sd.set_local_max_iters(m.local_max_iters)
        self._two_stage_on = False
        ppd = m.post_process_data
        for k in ppd:
            ppd[k] = []
        ppd["global_residual_vector_out"] = [[] for _ in range(P)]
        m.iter_count = 0

    def _announce_protocol(self):
        """Says once per run which exchange / termination protocol actually runs: several of the
        reference's flags select variants of MPI RMA that have no counterpart over RCCL and are
        accepted as aliases (INTEGRATION.md section 1 lists them)."""
        cs, cv = self.settings.comm_settings, self.settings.convergence_settings
        if not cs.enable_onesided:
            return
        notes = []
        if cs.enable_put or cs.enable_one_by_one or cs.enable_flush_local or cs.enable_lock_local \
                or not cs.enable_get or not cs.enable_flush_all or not cs.enable_lock_all:
            notes.append("RMA flavour flags (put/get, one_by_one, flush/lock) select no different code: "
                         "halos travel as RCCL send/recv pairs")
        if cs.enable_overlap:
            proto = getattr(self, "_termination", "flood")
            notes.append("termination protocol: " + {
                "flood": "decentralised flooding of (converged mask, stop iteration) over neighbour messages "
                         "(conv_tools.hpp:213-275)",
                "tree": "centralised tree: converged counts reduced to subdomain 0 along a binary tree, the "
                        "verdict broadcast down it (conv_tools.hpp:147-209)"}[proto])
            if cv.enable_accumulate:
                notes.append("enable_accumulate (MPI_Accumulate(MIN) of the norms) has no RCCL counterpart: "
                             "the flags travel with the halo messages")
        elif getattr(self, "_win", None) is not None:
            notes.append("free-running one-sided exchange: halo %s through peer-mapped device windows, no "
                         "matched receive, no collective in the loop; termination: %s" %
                         ("put" if cs.enable_put else "get",
                          "centralised tree (conv_tools.hpp:147-209)" if cv.enable_global_simple_tree else
                          ("decentralised, accumulated counters" if cv.enable_accumulate else
                           "decentralised flag propagation (conv_tools.hpp:213-275)")))
        else:
            notes.append("one-sided without enable_overlap on a communicator without node windows: local "
                         "tests, flags all-gathered per iteration (a host collective stands in for the "
                         "window reads)")
        if not cv.put_all_local_residual_norms:
            notes.append("put_all_local_residual_norms=false changes nothing: norms never travel in one-sided mode")
        for n in notes:
            self._print(" [schwz] " + n)

    def _setup_windows(self):
        """Communicate::setup_windows (source/communicate.cpp; include/communicate.hpp:67-224): every
        rank exposes its receive and its send buffer as a window the neighbours map, and learns at
        which offset of a neighbour's window its own values live (put_displacements /
        get_displacements, restricted_schwarz.cpp:624-658)."""
        s, be, comm = self.settings, self.backend, self.comm
        # HIP IPC handles and the POSIX shared-memory segment only mean something on ONE node
        import socket
        hosts = set(comm.share(socket.gethostname()))
        if len(hosts) > 1:
            raise capi.SchwzError(capi.ERR_INVALID,
                                  "the free-running one-sided mode maps the neighbours' device windows through HIP "
                                  "IPC and needs all ranks on one node; this job spans %d hosts (%s)" %
                                  (len(hosts), ", ".join(sorted(hosts))))
        me = comm.rank
        sd = self.subdomains[me]
        single = s.use_mixed_precision
        recv_w = be.window(max(sd.num_recv, 1), single)
        send_w = be.window(max(sd.num_send, 1), single)
        roff, soff = sd.recv_offsets(), sd.send_offsets()
        mine = dict(recv=recv_w.handle, send=send_w.handle,
                    recv_off={p: roff[k] for k, (p, _) in enumerate(sd.get_lists())},
                    send_off={q: soff[k] for k, (q, _) in enumerate(sd.put_lists())})
        everyone = comm.share(mine)
        peers_recv, peers_send = {}, {}
        for q, _ in sd.put_lists():   # "put": my values go into q's receive window, where q expects rank me
            peers_recv[q] = (be.open_window(everyone[q]["recv"]), everyone[q]["recv_off"][me])
        for p, _ in sd.get_lists():   # "get": p's values for me sit in p's send window
            peers_send[p] = (be.open_window(everyone[p]["send"]), everyone[p]["send_off"][me])
        self._win = dict(recv=recv_w, send=send_w, peers_recv=peers_recv, peers_send=peers_send, single=single,
                         host=comm.host_windows(), sent=np.zeros(comm.size, dtype=np.int32),
                         puts=[(q, len(ids)) for q, ids in sd.put_lists()],
                         gets=[(p, len(ids)) for p, ids in sd.get_lists()], roff=roff, soff=soff)
        comm.barrier()

    def close(self):
        """Releases the windows of the free-running one-sided mode in an order every rank can rely on:
        nobody iterates any more (barrier), the mappings of the neighbours' windows go, everybody has
        unmapped (barrier), the own exported buffers are freed, then the host windows (the creator unlinks
        the shared-memory segment).  Collective; called by a re-initialize and by the owner when done."""
        win = getattr(self, "_win", None)
        if win is None:
            return
        self._win = None
        comm = self.comm
        self.backend.synchronize()
        comm.barrier()
        for w, _ in list(win["peers_recv"].values()) + list(win["peers_send"].values()):
            w.close()
        comm.barrier()
        win["recv"].close()
        win["send"].close()
        win["host"] = None
        if hasattr(comm, "close_windows"):
            comm.close_windows()

    def _exchange_free_running(self, sd, stream):
        """exchange_boundary_onesided (restricted_schwarz.cpp:715-852): put = pack into the neighbours'
        receive windows, then scatter whatever MY receive window holds; get = pack into my send
        window, then scatter straight out of the neighbours' send windows.  Nobody waits for anybody."""
        w, cs = self._win, self.settings.comm_settings
        single = w["single"]
        if cs.enable_put:
            for k, (q, _) in enumerate(w["puts"]):
                win, off = w["peers_recv"][q]
                sd.pack_neighbor(k, win.at(off), single, stream)
            for k, (p, _) in enumerate(w["gets"]):
                sd.unpack_neighbor(k, w["recv"].at(w["roff"][k]), single, stream)
        else:
            for k, (q, _) in enumerate(w["puts"]):
                sd.pack_neighbor(k, w["send"].at(w["soff"][k]), single, stream)
            for k, (p, _) in enumerate(w["gets"]):
                win, off = w["peers_send"][p]
                sd.unpack_neighbor(k, win.at(off), single, stream)

    def _termination_free_running(self, me, sd, lres, converged_local):
        """check_global_convergence, one-sided branch (solve.cpp:876-943) on the shared-memory windows.
        Returns num_converged_procs."""
        cv, be, m = self.settings.convergence_settings, self.backend, self.metadata
        P, it = m.num_subdomains, m.iter_count
        hw = self._win["host"]
        resid, ppd = hw["resid"], m.post_process_data
        # window_residual_vector (conv_tools.hpp:56-142): my smallest local residual so far, to everybody
        # (put_all_local_residual_norms) or min-accumulated along the neighbour graph
        mine = resid[me]
        mine[me] = min(mine[me], lres)
        prev = ppd["global_residual_vector_out"][me][-1] if it > 0 and ppd["global_residual_vector_out"][me] else None
        if cv.put_all_local_residual_norms:
            if it > 0 and mine[me] != prev:
                for j in range(P):
                    if j != me:
                        resid[j][me] = mine[me]
        else:
            big = np.finfo(np.float64).max
            for q, cnt_q in self._win["puts"]:
                if cnt_q == 0:
                    continue
                for j in range(P):
                    if j != q and mine[j] != big:
                        be.atomic_min(resid[q], j, float(mine[j]))
        for j in range(P):
            ppd["global_residual_vector_out"][j].append(float(mine[j]))
        if cv.enable_global_simple_tree:
            # Yamazaki et al. 2019 (conv_tools.hpp:147-209): slots 0 / 1 = my children have reported, slot 2 =
            # the verdict coming down; a node reports once (slot 0 := 2), when its children have and it
            # passes its own test in this very iteration
            t = hw["tree"]
            c = t[me]
            if (((c[0] == 1 and c[1] == 1) or (c[0] == 1 and me == P // 2 - 1) or (me >= P // 2 and c[0] != 2))
                    and converged_local):
                if me == 0:
                    c[2] = 1
                else:
                    t[(me - 1) // 2][1 if me % 2 == 0 else 0] = 1
                c[0] = 2
            if c[2] == 1:
                for child in (2 * me + 1, 2 * me + 2):
                    if child < P:
                        t[child][2] = 1
                c[1] += 1
                return P
            return 0
        if cv.enable_accumulate:
            # conv_tools.hpp:229-246: +1 on everybody's counter.  The reference adds again in EVERY
            # iteration a rank passes its test, so its counters overshoot num_subdomains and the
            # "== num_subdomains" exit can be missed or hit early; here a rank adds once.
            cnt = hw["count"]
            if converged_local and not self._win.get("counted"):
                for j in range(P):
                    be.atomic_add(cnt, j, 1)
                self._win["counted"] = True
            return int(cnt[me])
        # decentralised flag propagation (conv_tools.hpp:247-273): what I know goes to my out-neighbours, once
        fl, sent = hw["flags"], self._win["sent"]
        if converged_local:
            fl[me][me] = 1
        local = fl[me].copy()
        for q, cnt_q in self._win["puts"]:
            if cnt_q == 0:
                continue
            for j in range(P):
                if sent[j] == 0 and local[j] == 1:
                    fl[q][j] = 1
        sent[:] = local
        return int(local.sum())

    def _step_free_running(self):
        """`enable_onesided` on a communicator with node windows: the reference's asynchronous
        iteration (schwarz_base.cpp:387-452 with exchange_boundary_onesided and the one-sided branch
        of check_global_convergence).  Every rank runs at its own pace: halos are whatever the
        neighbours' last put left in my window (or what their send window holds when I get), the
        local test feeds the tree / decentralised termination protocol on shared-memory windows, and
        a rank leaves the loop when ITS window says all have converged."""
        s, m, be, comm = self.settings, self.metadata, self.backend, self.comm
        cv = s.convergence_settings
        self._two_stage()
        me = comm.rank
        sd = self.subdomains[me]
        ppd, tol, it = m.post_process_data, m.tolerance, m.iter_count
        stream = be.stream()
        t0 = time.perf_counter()
        if it > 0:  # restricted_schwarz.cpp:725
            self._exchange_free_running(sd, stream)
        t1 = time.perf_counter()
        sd.update_boundary(stream)
        t2 = time.perf_counter()
        spec = self.speculative_solve and hasattr(sd, "check_and_solve_launch") and tol >= 0.0
        if spec:
            sd.check_and_solve_launch(stream)
            lres = sd.local_residual_wait()
        else:
            lres = sd.local_residual(stream) if tol >= 0.0 else -1.0
        if self._lres0[me] < 0.0:
            self._lres0[me] = lres
        if np.isnan(lres):
            raise capi.SchwzError(capi.ERR_DIVERGED, "local residual is NaN")
        ppd["local_residual_vector_out"].append(lres)
        m.current_residual_norm = lres
        m.min_residual_norm = lres if it == 0 else min(lres, m.min_residual_norm)
        iter_cond = ((it > m.max_iters * 0.05) or m.max_iters < 1000) \
            if cv.enable_global_check_iter_offset else True
        if tol > 0.0 and iter_cond:
            converged_local = _ratio(lres, self._lres0[me]) <= tol
            self._num_converged = self._termination_free_running(me, sd, lres, converged_local)
        t3 = time.perf_counter()
        tm = self._timings
        tm[0].append(t1 - t0)
        tm[1].append(t2 - t1)
        tm[2].append(t3 - t2)
        if self._num_converged == m.num_subdomains:
            return True
        if not spec:
            sd.local_solve(stream)
        self._log_local_solve([(me, sd)])
        t4 = time.perf_counter()
        sd.restrict(stream)
        tm[3].append(t4 - t3)
        tm[4].append(time.perf_counter() - t4)
        m.iter_count += 1
        return False

    def _step_overlapped(self):
        """`enable_onesided` + `enable_overlap`: the asynchronous flavour of the iteration as
        this build defines it (BASELINE config 5).  The halo exchange of iteration k is posted
        before the local solve of iteration k -- on a side stream under RCCL -- and consumed at
        the start of iteration k+1, so it is hidden behind the solve and halos are one iteration
        older than in the synchronous loop.  There is no global collective: every subdomain
        tests itself (solve.cpp:913-915) and floods (mask of converged subdomains, agreed stop
        iteration) to its neighbours with each message, the decentralised protocol of
        conv_tools.hpp:213-275 on matched messages; the first subdomain that sees the full mask
        at iteration k proposes stop = k + P, the minimum wins, and everybody stops together."""
        s, m, be, comm = self.settings, self.metadata, self.backend, self.comm
        self._two_stage()
        P = m.num_subdomains
        locals_ = list(self.subdomains.items())
        ppd = m.post_process_data
        tol = m.tolerance
        it = m.iter_count
        stream = be.stream()
        full = (1 << P) - 1
        t0 = time.perf_counter()
        # (a) consume what was posted one iteration ago
        if self._pending is not None:
            comm.finish_exchange(self._pending["halo"])
            for me, sd in locals_:
                self._unpack(me, sd, stream)
            for me, msgs in comm.finish_flags(self._pending["flags"]).items():
                for mk, st in msgs:
                    self._mask[me] |= mk
                    self._stop[me] = min(self._stop[me], st)
            self._pending = None
        last = it == m.max_iters - 1 or any(self._stop[me] == it for me, _ in locals_)
        # (b) post this iteration's halos: x~ after the previous restriction
        halo = None
        if not last:
            for me, sd in locals_:
                self._pack(me, sd, stream)
            halo = comm.start_exchange(self._sends, self._recvs, overlap=True)
        t1 = time.perf_counter()
        # (c) boundary update, local test + local solve (enqueued together), restriction
        for _, sd in locals_:
            sd.update_boundary(stream)
        t2 = time.perf_counter()
        fused = hasattr(locals_[0][1], "check_and_solve_launch") and tol >= 0.0
        if fused:
            for _, sd in locals_:
                sd.check_and_solve_launch(stream)
        for me, sd in locals_:
            lres = -1.0
            if tol >= 0.0:
                lres = sd.local_residual_wait() if fused else sd.local_residual(stream)
                if self._lres0[me] < 0.0:
                    self._lres0[me] = lres
            if np.isnan(lres):
                raise capi.SchwzError(capi.ERR_DIVERGED, "local residual is NaN")
            ppd["local_residual_vector_out"].append(lres)
            m.current_residual_norm = lres
            if tol > 0.0 and _ratio(lres, self._lres0[me]) <= tol:
                self._mask[me] |= 1 << me
            if self._mask[me] == full and self._stop[me] == NEVER:
                self._stop[me] = it + P
        if not last:
            neighbours = self._neighbour_ranks
            flags = comm.start_flags({me: (self._mask[me], self._stop[me]) for me, _ in locals_},
                                     neighbours)
            self._pending = dict(halo=halo, flags=flags)
        t3 = time.perf_counter()
        tm = self._timings
        tm[0].append(t1 - t0)
        tm[1].append(t2 - t1)
        tm[2].append(t3 - t2)
        if any(self._stop[me] == it for me, _ in locals_):
            self._num_converged = P
            return True
        if not fused:
            for _, sd in locals_:
                sd.local_solve(stream)
        self._log_local_solve(locals_)
        t4 = time.perf_counter()
        for _, sd in locals_:
            sd.restrict(stream)
        tm[3].append(t4 - t3)
        tm[4].append(time.perf_counter() - t4)
        m.iter_count += 1
        return False

    def step(self):
        """One pass of the loop body of SchwarzBase::run (schwarz_base.cpp:387-452).
        Returns True when the convergence test fired (no local solve is done then)."""
        s, m, be, comm = self.settings, self.metadata, self.backend, self.comm
        cs, cv = s.comm_settings, s.convergence_settings
        if cs.enable_onesided and cs.enable_overlap:
            return self._step_overlapped()
        if cs.enable_onesided and getattr(self, "_win", None) is not None:
            return self._step_free_running()
        self._two_stage()
        P = m.num_subdomains
        locals_ = list(self.subdomains.items())
        ppd = m.post_process_data
        tol = m.tolerance
        it = m.iter_count
        stream = be.stream()
        t0 = time.perf_counter()
        # 0 boundary exchange (one-sided mode skips iteration 0, restricted_schwarz.cpp:725)
        if not (cs.enable_onesided and it == 0):
            self._exchange()
        t1 = time.perf_counter()
        # 1 boundary update
        for _, sd in locals_:
            sd.update_boundary(stream)
        t2 = time.perf_counter()
        # 2 convergence check (solve.cpp:959-1005).  The residual kernels are enqueued, then
        # the local solve (step 3) is enqueued right behind them BEFORE the host waits for
        # the 8-byte norm: the GPU keeps working while the host runs the global check.  The
        # solve only writes y (and CG work vectors); if the verdict is "converged" its result
        # is simply not written back (step 4 is skipped), exactly like the reference.
        spec = self.speculative_solve and hasattr(locals_[0][1], "local_residual_launch")
        if tol >= 0.0 and spec:
            for _, sd in locals_:
                sd.check_and_solve_launch(stream)
        iter_cond = ((it > m.max_iters * 0.05) or m.max_iters < 1000) \
            if cv.enable_global_check_iter_offset else True
        # the P norms of the global test as a device-side all-gather (RCCL), enqueued on a side stream behind the
        # norm's own event: it runs beside the local solve, and the host then waits for ONE event instead of its own
        # scalar plus a gloo collective (every rank takes this branch or none: the conditions are global settings)
        norm_handle = None
        if (tol > 0.0 and spec and iter_cond and cv.enable_global_check and not cs.enable_onesided
                and getattr(comm, "device_norms", False) and len(locals_) == 1):
            sd0 = locals_[0][1]
            try:
                norm_handle = comm.start_allgather_norm_sq(lambda ptr, raw: sd0.norm_sq_to_device(ptr, raw))
            except Exception as e:  # e.g. the second RCCL communicator cannot be brought up: every rank sees that
                if getattr(comm, "_norm_used", False):
                    raise
                import warnings
                warnings.warn("device-side all-gather of the residual norms failed at its first use (%s: %s); "
                              "the gloo group takes over" % (type(e).__name__, e))
                comm.device_norms = False
                norm_handle = None
            else:
                comm._norm_used = True
        lres = {}
        for me, sd in locals_:
            if tol < 0.0:
                lres[me] = -1.0
            elif spec:
                lres[me] = sd.local_residual_wait()
            else:
                lres[me] = sd.local_residual(stream)
            if self._lres0[me] < 0.0:
                self._lres0[me] = lres[me]
            if np.isnan(lres[me]):
                raise capi.SchwzError(capi.ERR_DIVERGED, "local residual is NaN")
            ppd["local_residual_vector_out"].append(lres[me])
            ppd["local_converged_resnorm"].append(
                lres[me] / self._lres0[me] if self._lres0[me] != 0 else float("nan"))
            m.current_residual_norm = lres[me]
            m.min_residual_norm = lres[me] if it == 0 else min(lres[me], m.min_residual_norm)
        if tol > 0.0 and iter_cond:
            if cv.enable_global_check and not cs.enable_onesided:
                if norm_handle is not None:  # solve.cpp:890-891 on the device
                    allres = [float(np.sqrt(v)) for v in comm.finish_allgather_norm_sq(norm_handle)]
                else:
                    allres = comm.allgather_scalars(lres)  # solve.cpp:890-891
                gres = 0.0
                for j in range(P):
                    ppd["global_residual_vector_out"][j].append(allres[j])
                    gres += allres[j]  # SUM of the norms (solve.cpp:895-905)
                self._gres = gres
                if self._gres0 < 0.0:
                    self._gres0 = gres
                self._num_converged = P if _ratio(gres, self._gres0) <= tol else 0
            elif cs.enable_onesided:
                # local test (solve.cpp:913-915) + monotone flags (conv_tools.hpp:249-251)
                for me, _ in locals_:
                    if _ratio(lres[me], self._lres0[me]) <= tol:
                        self._flags[me] = True
                allflags = comm.allgather_scalars(
                    {me: float(self._flags[me]) for me, _ in locals_})
                self._num_converged = int(sum(allflags))
            else:
                self._num_converged = 0  # never converges on this branch
        t3 = time.perf_counter()
        if np.isnan(self._gres) or self._gres > 1e12:
            raise capi.SchwzError(capi.ERR_DIVERGED,
                                  " Rank %d diverged in %d iters " % (comm.rank, it))
        tm = self._timings
        tm[0].append(t1 - t0)
        tm[1].append(t2 - t1)
        tm[2].append(t3 - t2)
        if self._num_converged == P:
            return True
        # 3 local solve (already enqueued above in speculative mode)
        if not (tol >= 0.0 and spec):
            for _, sd in locals_:
                sd.local_solve(stream)
        self._log_local_solve(locals_)
        # step 0 of the NEXT iteration, posted now: pack + send / recv on a side stream as soon as the
        # solves have finalised their boundary rows, beside the rest of the solution update and step 4
        if not cs.enable_onesided and self._early_exchange_ok():
            self._post_early_exchange()
        t4 = time.perf_counter()
        # 4 restricted write-back
        for _, sd in locals_:
            sd.restrict(stream)
        t5 = time.perf_counter()
        tm[3].append(t4 - t3)
        tm[4].append(t5 - t4)
        m.iter_count += 1
        return False

    def finish_run(self, elapsed, gather_solution=True):
        """The tail of SchwarzBase::run (schwarz_base.cpp:453-503)."""
        m, comm = self.metadata, self.comm
        P = m.num_subdomains
        stream = self.backend.stream()
        locals_ = list(self.subdomains.items())
        m.time_struct = [(i, comm.rank, len(self._timings[i]), TIMING_NAMES[i], self._timings[i])
                         for i in range(5)]
        converged = self._num_converged == P
        out = dict(iter_count=m.iter_count, converged=converged, elapsed=elapsed,
                   residual_norm=None, rhs_norm=None, sol_norm=None, solution=None)
        if not converged:
            self._print("Rank %d did not converge in %d iterations." % (comm.rank, m.iter_count))
        else:
            self._print(" Rank %d converged in %d iterations " % (comm.rank, m.iter_count))
        # Solve::compute_residual_norm (solve.cpp:1025-1085): the interiors form the
        # solution; (A x) on the interior rows needs fresh overlap values, hence
        # one more exchange.
        self._exchange()
        parts = {me: sd.true_residual_sq(stream) for me, sd in locals_}
        res_sq = sum(comm.allgather_scalars(parts))
        rhs_sq = sum(comm.allgather_scalars(
            {me: float(np.sum(self._rhs_first_rows(sd, sd.local_size) ** 2))
             for me, sd in locals_}))
        out["residual_norm"] = float(np.sqrt(res_sq))
        out["rhs_norm"] = float(np.sqrt(rhs_sq))
        if gather_solution:
            pieces = {me: sd.get_interior(stream) for me, sd in locals_}
            sol_sq = sum(comm.allgather_scalars({me: float(np.dot(pieces[me], pieces[me]))
                                                 for me in pieces}))
            out["sol_norm"] = float(np.sqrt(sol_sq))
            out["solution"] = comm.gather_vectors(pieces)
        if converged:
            self._print(" residual norm %g\n relative residual norm of solution %g\n"
                        " Time taken for solve %g" %
                        (out["residual_norm"], out["residual_norm"] / out["rhs_norm"], elapsed))
        self.result = out
        return out

    def run(self, gather_solution=True):
        """SchwarzBase::run (schwarz_base.cpp:323-506).  Returns a dict with the
        solution (on the root rank) and the run statistics."""
        m, be, comm = self.metadata, self.backend, self.comm
        self.begin_run()
        be.synchronize()
        comm.barrier()
        start = time.perf_counter()
        while m.iter_count < m.max_iters:
            if self.step():
                break
        be.synchronize()
        comm.barrier()  # schwarz_base.cpp:453
        elapsed = time.perf_counter() - start
        return self.finish_run(elapsed, gather_solution)
