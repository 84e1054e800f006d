"""Multi-process CPU tests of the N>1 host path (one subdomain per process over
torch.distributed/gloo): the index handshake, the grouped halo send/recv, the all-gathered
global convergence rule and the solution gather of schwz_amd.SolverRAS + TorchDistComm.
Compute is done by the test-only oracle backend, so the distributed run must reproduce the
oracle's single-process lockstep run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    for p in (HERE, os.path.join(os.path.dirname(HERE), "schwarz-lib_amd"),
              os.path.join(os.path.dirname(HERE), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import schwz_amd as S
    from oracle_backend import OracleBackend
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = S.Settings(**case["settings"])
        if case.get("onesided"):
            s.comm_settings.enable_onesided = True
            s.convergence_settings.enable_global_simple_tree = True
        if case.get("overlap"):
            s.comm_settings.enable_overlap = True
            s.convergence_settings.enable_global_simple_tree = False
            s.convergence_settings.enable_decentralized_leader_election = True
        m = S.Metadata(**case["metadata"])
        solver = S.SolverRAS(s, m, comm=S.TorchDistComm(), backend=OracleBackend(), quiet=True)
        assert m.num_subdomains == world and m.my_rank == rank
        solver.initialize()
        out = solver.run()
        if rank == 0:
            hist = np.array(m.post_process_data["global_residual_vector_out"])
            np.savez(out_path, solution=out["solution"], iter_count=out["iter_count"],
                     converged=out["converged"], residual_norm=out["residual_norm"],
                     rhs_norm=out["rhs_norm"], hist=hist, first_row=np.asarray(m.first_row))
        else:
            assert out["solution"] is None
    finally:
        dist.destroy_process_group()


def _run(world, case, tmp_path):
    out_path = str(tmp_path / "out.npz")
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, out_path)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.terminate()
        assert p.exitcode == 0, "worker failed with exit code %s" % p.exitcode
    return np.load(out_path)


CASES = {
    "lap2d_cg_jacobi": dict(
        world=2, settings=dict(),
        metadata=dict(oned_laplacian_size=20, tolerance=1e-8, max_iters=300,
                      local_precond="block-jacobi", precond_max_block_size=1)),
    "lap3d_truncated_cg": dict(
        world=3, settings=dict(laplacian_dim=3, laplacian_shape=(6, 5, 9)),
        metadata=dict(tolerance=1e-6, max_iters=300, local_precond="block-jacobi",
                      precond_max_block_size=1, local_solver_tolerance=0.0, local_max_iters=5)),
    "lap2d_cg_ilu": dict(
        world=2, settings=dict(),
        metadata=dict(oned_laplacian_size=18, tolerance=1e-8, max_iters=300, local_precond="ilu")),
    "lap2d_cg_isai": dict(
        world=2, settings=dict(),
        metadata=dict(oned_laplacian_size=18, tolerance=1e-8, max_iters=300, local_precond="isai")),
    "lap2d_cg_block_jacobi8": dict(
        world=3, settings=dict(),
        metadata=dict(oned_laplacian_size=18, tolerance=1e-8, max_iters=300,
                      local_precond="block-jacobi", precond_max_block_size=8)),
    "lap2d_two_stage": dict(
        world=2, settings=dict(reset_local_crit_iter=3),
        metadata=dict(oned_laplacian_size=18, tolerance=1e-8, max_iters=300, local_precond="block-jacobi",
                      precond_max_block_size=1, local_solver_tolerance=1e-10, local_max_iters=2,
                      updated_max_iters=40)),
    # eight ranks, the rank count of the multi-GPU benchmark: slab partition, two neighbours each
    "lap3d_eight_slabs": dict(
        world=8, settings=dict(laplacian_dim=3, laplacian_shape=(6, 5, 32)),
        metadata=dict(tolerance=1e-6, max_iters=400, local_precond="block-jacobi",
                      precond_max_block_size=1, local_solver_tolerance=0.0, local_max_iters=6)),
    "lap3d_eight_slabs_overlapped": dict(
        world=8, settings=dict(laplacian_dim=3, laplacian_shape=(5, 4, 32)), onesided=True, overlap=True,
        metadata=dict(tolerance=1e-5, max_iters=600, local_precond="block-jacobi",
                      precond_max_block_size=1, local_solver_tolerance=0.0, local_max_iters=5)),
    "lap2d_direct_overlap3": dict(
        world=2, settings=dict(local_solver="direct-ginkgo", overlap=3),
        metadata=dict(oned_laplacian_size=16, tolerance=1e-9, max_iters=300)),
    "lap2d_onesided": dict(
        world=2, settings=dict(), onesided=True,
        metadata=dict(oned_laplacian_size=16, tolerance=1e-6, max_iters=300)),
    # asynchronous flavour: halos one iteration late, decentralised stop agreement
    "lap2d_overlapped_decentralized": dict(
        world=3, settings=dict(), onesided=True, overlap=True,
        metadata=dict(oned_laplacian_size=18, tolerance=1e-6, max_iters=400)),
    "lap3d_overlapped_truncated_cg": dict(
        world=4, settings=dict(laplacian_dim=3, laplacian_shape=(5, 4, 12)), onesided=True, overlap=True,
        metadata=dict(tolerance=1e-5, max_iters=400, local_precond="block-jacobi",
                      precond_max_block_size=1, local_solver_tolerance=0.0, local_max_iters=4)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_distributed_run_reproduces_lockstep_oracle(oracle, name, tmp_path):
    case = dict(CASES[name])
    world = case.pop("world")
    got = _run(world, case, tmp_path)
    st, md = case["settings"], case["metadata"]
    if st.get("laplacian_dim") == 3:
        rp, col, val = oracle.laplacian3d(*st["laplacian_shape"])
    else:
        rp, col, val = oracle.laplacian2d(md["oned_laplacian_size"])
    N = len(rp) - 1
    s = oracle.make_settings(
        max_iters=md["max_iters"], tol=md["tolerance"], overlap=st.get("overlap", 2),
        local_solver=oracle.SOLVER_DIRECT if st.get("local_solver", "").startswith("direct") else 0,
        precond=oracle.precond_code(md.get("local_precond"), md.get("precond_max_block_size", 1))[0],
        precond_block_size=oracle.precond_code(md.get("local_precond"), md.get("precond_max_block_size", 1))[1],
        local_tol=md.get("local_solver_tolerance", 1e-12), local_max_iters=md.get("local_max_iters", -1),
        enable_onesided=int(bool(case.get("onesided"))), enable_overlap=int(bool(case.get("overlap"))),
        reset_local_crit_iter=st.get("reset_local_crit_iter", -1), updated_max_iters=md.get("updated_max_iters", -1))
    fr = oracle.first_rows_regular(N, world)
    assert np.array_equal(got["first_row"], fr)
    ref = oracle.ras_run(rp, col, val, np.ones(N), world, fr, s)
    assert bool(got["converged"]) and ref["converged"]
    assert int(got["iter_count"]) == ref["iter_count"]
    # same arithmetic in the same order: bit-identical
    assert np.array_equal(got["solution"], ref["solution"])
    if not case.get("onesided"):
        assert np.array_equal(got["hist"].sum(axis=0), ref["hist_global"])
        assert np.array_equal(got["hist"].T, ref["hist_local"])
    assert abs(float(got["residual_norm"]) - ref["residual_norm"]) <= 1e-10 * ref["rhs_norm"]
    assert abs(float(got["rhs_norm"]) - ref["rhs_norm"]) <= 1e-12 * ref["rhs_norm"]


# ---- the free-running one-sided mode: halo windows mapped by the neighbours, no collective in the loop ----

def _free_worker(rank, world, port, case, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    for p in (HERE, os.path.join(os.path.dirname(HERE), "schwarz-lib_amd"),
              os.path.join(os.path.dirname(HERE), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import time
    import torch.distributed as dist
    import schwz_amd as S
    from oracle_backend import OracleBackend
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = S.WindowComm()
    try:
        s = S.Settings(**case["settings"])
        s.comm_settings.enable_onesided = True
        s.comm_settings.enable_put = case["put"]
        s.comm_settings.enable_get = not case["put"]
        cv = s.convergence_settings
        cv.enable_global_simple_tree = case["protocol"] == "tree"
        cv.enable_decentralized_leader_election = case["protocol"] != "tree"
        cv.enable_accumulate = case["protocol"] == "accumulate"
        cv.put_all_local_residual_norms = case.get("put_all", True)
        m = S.Metadata(**case["metadata"])
        solver = S.SolverRAS(s, m, comm=comm, backend=OracleBackend(), quiet=True)
        solver.initialize()
        if case.get("reinit"):
            # a second initialize() closes the windows of the first one (peers unmapped before the owners
            # free, the shared-memory segment unlinked) and sets up new ones
            first = comm._shm.name
            solver.initialize()
            assert comm._shm.name != first and not os.path.exists("/dev/shm/" + first.lstrip("/"))
        if case.get("slow_rank") == rank:
            # one rank runs at a third of the others' pace: nobody may wait for it inside the loop
            step = solver.step

            def slow_step():
                time.sleep(0.004)
                return step()
            solver.step = slow_step
        out = solver.run()
        hist = np.array(m.post_process_data["global_residual_vector_out"])
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), iter_count=out["iter_count"],
                 converged=out["converged"], residual_norm=out["residual_norm"], rhs_norm=out["rhs_norm"],
                 solution=out["solution"] if rank == 0 else np.zeros(0), hist=hist,
                 local=np.array(m.post_process_data["local_residual_vector_out"]))
        solver.close()
        assert solver._win is None and comm._shm is None
    finally:
        comm.close_windows()
        dist.destroy_process_group()


FREE_CASES = {
    "put_tree": dict(world=3, put=True, protocol="tree", reinit=True),
    "get_tree": dict(world=4, put=False, protocol="tree"),
    "put_decentralized": dict(world=3, put=True, protocol="decentralized"),
    "get_decentralized_propagated_norms": dict(world=4, put=False, protocol="decentralized", put_all=False),
    "put_accumulate": dict(world=3, put=True, protocol="accumulate"),
    "put_tree_one_slow_rank": dict(world=4, put=True, protocol="tree", slow_rank=2),
    "get_decentralized_one_slow_rank": dict(world=3, put=False, protocol="decentralized", slow_rank=0),
}


@pytest.mark.parametrize("name", sorted(FREE_CASES))
def test_free_running_onesided_mode(oracle, name, tmp_path):
    """enable_onesided on a communicator with node windows (schwz_amd.WindowComm): every rank iterates
    at its own pace, halo values are put into / got from windows the neighbours map, termination runs
    on shared-memory windows (tree, decentralised flags, accumulated counters).  The run is not
    deterministic (that is its point), so it is checked by properties: every rank leaves the loop
    converged, iteration counts may differ between ranks but all pass their local test at the end,
    the assembled solution solves the system to the accuracy the tolerance implies, and a rank held
    back by sleeps does not hold the others back inside the loop."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as sl
    case = dict(FREE_CASES[name])
    world = case.pop("world")
    n = 18
    case["settings"] = dict()
    case["metadata"] = dict(oned_laplacian_size=n, tolerance=1e-7, max_iters=4000)
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_free_worker, args=(r, world, port, case, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    for p in procs:
        if p.is_alive():
            p.terminate()
        assert p.exitcode == 0, "worker failed with exit code %s" % p.exitcode
    res = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    assert all(bool(r["converged"]) for r in res)
    iters = [int(r["iter_count"]) for r in res]
    assert min(iters) > 5 and max(iters) < 4000
    for r in res:   # every rank has passed its own test at some point (flags and tree reports are sticky)
        assert r["local"].min() <= 1e-7 * r["local"][0]
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    A = sp.csr_matrix((val, col, rp), shape=(N, N))
    x_ref = sl.spsolve(A.tocsc(), np.ones(N))
    x = res[0]["solution"]
    assert x.shape == (N,)
    assert np.abs(x - x_ref).max() <= 1e-4 * np.abs(x_ref).max()
    assert float(res[0]["residual_norm"]) / float(res[0]["rhs_norm"]) < 1e-4
    if case.get("slow_rank") is not None:
        slow = case["slow_rank"]
        others = [iters[r] for r in range(world) if r != slow]
        # the fast ranks ran ahead: they did clearly more iterations than the rank that sleeps
        assert min(others) > iters[slow]


# ---------------------------------------------------------------------------------------------------
# Distributed ingest of a Matrix-Market file: the root parses and partitions, every rank
# receives the rows its subdomain reads and sets its subdomain up on that part alone.
# ---------------------------------------------------------------------------------------------------

class _HostBackend:
    """What SolverRAS needs from a backend up to the subdomain index sets: the host-side entry points of the
    product library (no GPU involved)."""

    def __init__(self):
        import schwz_amd.core as core
        self.problem_from_matrix_market = core.Problem.from_matrix_market
        self.problem_from_rows = core.Problem.from_rows
        self.partition_regular = core.partition_regular
        self.partition_regular2d = core.partition_regular2d


def _ingest_worker(rank, world, port, path, partition, overlap, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    for p in (HERE, os.path.join(os.path.dirname(HERE), "schwarz-lib_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import schwz_amd as S
    import schwz_amd.core as core
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = S.Settings(matrix_filename=path, partition=partition, overlap=overlap, explicit_laplacian=False)
        m = S.Metadata()
        solver = S.SolverRAS(s, m, comm=S.TorchDistComm(), backend=_HostBackend(), quiet=True)
        assert solver._distributed_ingest()
        part = solver._ingest_distributed()
        # this rank's subdomain from ITS part of the matrix ...
        sd = core.Subdomain(part, world, rank, overlap, m.first_row)
        # ... and from the whole file, parsed here only to check: same partition, same permutation
        ref_solver = S.SolverRAS(S.Settings(matrix_filename=path, partition=partition, overlap=overlap,
                                            explicit_laplacian=False), S.Metadata(),
                                 comm=S.InProcessComm(world), backend=_HostBackend(), quiet=True)
        full = ref_solver._partition(core.Problem.from_matrix_market(path))
        ref = core.Subdomain(full, world, rank, overlap, ref_solver.metadata.first_row)
        assert np.array_equal(np.asarray(m.first_row), np.asarray(ref_solver.metadata.first_row))
        if ref_solver.metadata.permutation is None:
            assert m.permutation is None
        else:
            assert np.array_equal(m.permutation, ref_solver.metadata.permutation)
        assert part.N == full.N and part.nnz < full.nnz  # it really holds a part only
        for a, b in zip(sd.local_matrix() + sd.interface_matrix(), ref.local_matrix() + ref.interface_matrix()):
            assert np.array_equal(a, b)
        assert np.array_equal(sd.local_to_global, ref.local_to_global)
        assert [(q, list(ids)) for q, ids in sd.get_lists()] == [(q, list(ids)) for q, ids in ref.get_lists()]
        # a row this rank was not sent is an error, not a silent empty row
        other = int(ref_solver.metadata.first_row[(rank + world // 2 + 1) % world])
        if other not in set(sd.local_to_global[:sd.local_size_x].tolist()):
            cols, _ = part.row(other)
            try:
                core.Subdomain(part, world, (rank + world // 2 + 1) % world, overlap, m.first_row)
                raise AssertionError("a subdomain was built from rows this rank does not hold")
            except S.capi.SchwzError:
                pass
        open(os.path.join(out_dir, "ok_%d" % rank), "w").write("%d %d" % (part.nnz, full.nnz))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("partition,overlap", [("regular", 2), ("metis", 2), ("metis", 3)])
def test_distributed_ingest_builds_the_same_subdomains(tmp_path, partition, overlap):
    """One rank parses the Matrix-Market file and partitions; every rank receives the interior and overlap rows
    of its subdomain only (schwz_problem_extract_rows / _from_rows over the gloo host group) and builds index
    sets, local and interface matrices and get lists from that part: bit-identical to the subdomain built
    from the whole matrix, with the same first_row and permutation on every rank."""
    rng = np.random.default_rng(5)
    n = 23  # a 2-D grid graph with a few random long-range couplings, unsymmetric values
    N = n * n
    entries = {}
    for i in range(N):
        entries[(i, i)] = 4.0 + rng.random()
        for j in (i - 1 if i % n else -1, i + 1 if (i + 1) % n else -1, i - n, i + n):
            if 0 <= j < N:
                entries[(i, j)] = -1.0 - 0.1 * rng.random()
    for _ in range(40):
        i, j = int(rng.integers(N)), int(rng.integers(N))
        entries[(i, j)] = entries.get((i, j), 0.0) + 0.01
        entries[(j, i)] = entries.get((j, i), 0.0) + 0.02
    path = str(tmp_path / "graph.mtx")
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (N, N, len(entries)))
        for (i, j), v in entries.items():
            f.write("%d %d %.17g\n" % (i + 1, j + 1, v))
    world = 3
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_ingest_worker, args=(r, world, port, path, partition, overlap, str(tmp_path)))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    for p in procs:
        if p.is_alive():
            p.terminate()
        assert p.exitcode == 0, "worker failed with exit code %s" % p.exitcode
    assert all(os.path.exists(str(tmp_path / ("ok_%d" % r))) for r in range(world))
