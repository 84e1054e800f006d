"""Drop-in check: the reference's own driver benchmarking/bench_ras.cpp, compiled
UNCHANGED against schwarz-lib_amd/host/include (+ the gflags/ginkgo shims) and linked with
libschwz.so / libschwz_hip.so, run under mpiexec with --executor=hip.  The binary is built in
this repo's build/ directory where the reference checkout exists (`make -C schwarz-lib_amd
bench_ras`); it is a build artefact like the .so files and is never committed."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "schwarz-lib_amd", "build", "bench_ras")
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"


def _run(nranks, *flags, cwd=None, env=None):
    if not os.path.exists(BIN):
        pytest.skip("bench_ras binary not built (needs the reference checkout at build time)")
    if not os.path.exists(MPIEXEC):
        pytest.skip("no mpiexec on this machine")
    cmd = [MPIEXEC, "-n", str(nranks), BIN, "--executor=hip"] + list(flags)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=cwd,
                       env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


@pytest.mark.parametrize("nranks", [1, 2, 4])
def test_bench_ras_2d_laplacian_matches_oracle(oracle, nranks, tmp_path):
    n = 32
    out = _run(nranks, "--explicit_laplacian", "--set_1d_laplacian_size=%d" % n, "--enable_global_check",
               "--num_iters=500", "--set_tol=1e-8", "--timings_file=%s" % (tmp_path / "t"),
               "--write_comm_data", "--write_iters_and_residuals", cwd=str(tmp_path))
    iters = sorted(set(int(x) for x in re.findall(r"converged in (\d+) iterations", out)))
    assert len(iters) == 1, out
    rel = float(re.search(r"relative residual norm of solution ([0-9.eE+-]+)", out).group(1))
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    ref = oracle.ras_run(rp, col, val, np.ones(N), nranks, oracle.first_rows_regular(N, nranks),
                         oracle.make_settings(max_iters=500, tol=1e-8))
    assert ref["converged"] and iters[0] == ref["iter_count"]
    assert abs(rel - ref["residual_norm"] / ref["rhs_norm"]) <= 1e-6 * max(rel, 1e-12) + 1e-12
    # the driver's own CSV writers consumed time_struct / comm_data_struct
    t = (tmp_path / "t_00.csv").read_text().splitlines()
    assert t[0] == "func,total,avg,min,med,max"
    assert [l.split(",")[0] for l in t[1:]] == ["boundary_exchange", "boundary_update", "convergence_check",
                                                "local_solve", "expand_local_vec", "other"]
    # schwarz_base.cpp:456-472: one history file per rank, one line per outer iteration
    hist = (tmp_path / "iter_res_00.csv").read_text().splitlines()
    assert hist[0] == "iter,resnorm,localiter,localresnorm,timestamp"
    assert len(hist) == 1 + ref["iter_count"] + 1  # the converged iteration still records its residual
    assert float(hist[-1].split(",")[1]) < float(hist[1].split(",")[1])
    if nranks > 1:
        send = (tmp_path / "num_send_00.csv").read_text().splitlines()
        assert send[0].startswith("subdomain 0 has 1 neighbors") and send[1] == "my_id,to_id,num_send"
        assert "0,1,64" in send  # two grid lines of 32 values go to rank 1


def test_bench_ras_direct_solver_and_3d_extension(oracle):
    out = _run(2, "--matrix_filename=poisson3d:12x10x8", "--enable_global_check", "--num_iters=300",
               "--set_tol=1e-7", "--local_solver=direct-ginkgo", "--factor_ordering_natural")
    iters = set(int(x) for x in re.findall(r"converged in (\d+) iterations", out))
    rp, col, val = oracle.laplacian3d(12, 10, 8)
    N = len(rp) - 1
    ref = oracle.ras_run(rp, col, val, np.ones(N), 2, oracle.first_rows_regular(N, 2),
                         oracle.make_settings(max_iters=300, tol=1e-7, local_solver=oracle.SOLVER_DIRECT,
                                              natural_factor_ordering=1))
    assert iters == {ref["iter_count"]}, out


def test_bench_ras_refuses_cpu_executors():
    if not os.path.exists(BIN) or not os.path.exists(MPIEXEC):
        pytest.skip("bench_ras binary / mpiexec missing")
    p = subprocess.run([MPIEXEC, "-n", "1", BIN, "--executor=reference", "--explicit_laplacian"],
                       capture_output=True, text=True, timeout=120)
    assert "is not implemented" in p.stderr + p.stdout


def test_bench_ras_preconditioner_flags(oracle):
    """--local_precond=ilu and block-jacobi with --precond_max_block_size through the unchanged
    driver: same outer iteration counts as the oracle."""
    n = 30
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    for flags, (pc, bs) in ((("--local_precond=ilu",), (3, 1)),
                            (("--local_precond=block-jacobi", "--precond_max_block_size=8"), (2, 8))):
        out = _run(2, "--explicit_laplacian", "--set_1d_laplacian_size=%d" % n, "--enable_global_check",
                   "--num_iters=500", "--set_tol=1e-8", *flags)
        iters = set(int(x) for x in re.findall(r"converged in (\d+) iterations", out))
        ref = oracle.ras_run(rp, col, val, np.ones(N), 2, oracle.first_rows_regular(N, 2),
                             oracle.make_settings(max_iters=500, tol=1e-8, precond=pc, precond_block_size=bs))
        assert ref["converged"] and iters == {ref["iter_count"]}, out


def test_bench_ras_non_symmetric_gmres(oracle, convdiff, tmp_path):
    """--non_symmetric_matrix --restart_iter=m on a Matrix-Market input (bench_ras.cpp:115-116)."""
    rp, col, val = convdiff(26)
    N = len(rp) - 1
    path = str(tmp_path / "cd.mtx")
    rows = np.repeat(np.arange(N), np.diff(rp))
    with open(path, "w") as f:
        f.write("%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (N, N, rp[-1]))
        for r, c, v in zip(rows, col, val):
            f.write("%d %d %.17g\n" % (r + 1, c + 1, v))
    out = _run(2, "--matrix_filename=%s" % path, "--enable_global_check", "--num_iters=500", "--set_tol=1e-8",
               "--non_symmetric_matrix", "--restart_iter=12", "--local_precond=block-jacobi",
               "--precond_max_block_size=1", "--local_tol=1e-11", "--local_max_iters=500")
    iters = sorted(set(int(x) for x in re.findall(r"converged in (\d+) iterations", out)))
    ref = oracle.ras_run(rp, col, val, np.ones(N), 2, oracle.first_rows_regular(N, 2),
                         oracle.make_settings(max_iters=500, tol=1e-8, precond=1, local_tol=1e-11,
                                              local_max_iters=500, non_symmetric=1, restart_iter=12))
    assert ref["converged"] and len(iters) == 1 and abs(iters[0] - ref["iter_count"]) <= 1, out
    rel = float(re.search(r"relative residual norm of solution ([0-9.eE+-]+)", out).group(1))
    assert rel <= 1e-6


def test_bench_ras_halo_transport_selection(oracle):
    """Halos go device to device over RCCL when every rank owns a GPU; ranks sharing the one GPU
    of this box fall back to host staging over MPI, and a single rank can be forced onto the RCCL
    path (communicator set-up, grouped launch with no peers)."""
    n = 24
    base = ("--explicit_laplacian", "--set_1d_laplacian_size=%d" % n, "--enable_global_check", "--num_iters=400",
            "--set_tol=1e-8")
    rp, col, val = oracle.laplacian2d(n)
    N = n * n

    def iters_of(out):
        return set(int(x) for x in re.findall(r"converged in (\d+) iterations", out))

    out = _run(2, *base)
    assert "staged through host over MPI (ranks share a GPU)" in out
    ref2 = oracle.ras_run(rp, col, val, np.ones(N), 2, oracle.first_rows_regular(N, 2),
                          oracle.make_settings(max_iters=400, tol=1e-8))
    assert iters_of(out) == {ref2["iter_count"]}
    out = _run(1, *base, env={"SCHWZ_HALO": "rccl"})
    assert "RCCL send/recv, device to device" in out
    ref1 = oracle.ras_run(rp, col, val, np.ones(N), 1, oracle.first_rows_regular(N, 1),
                          oracle.make_settings(max_iters=400, tol=1e-8))
    assert iters_of(out) == {ref1["iter_count"]}
    out = _run(1, *base, env={"SCHWZ_HALO": "mpi"})
    assert "staged through host over MPI" in out and iters_of(out) == {ref1["iter_count"]}


@pytest.mark.parametrize("nranks", [2, 3])
def test_bench_ras_overlapped_onesided_matches_oracle(oracle, nranks):
    """--enable_onesided --enable_comm_overlap --global_convergence_type=decentralized: halos one
    iteration late, decentralised stop agreement (BASELINE config 5 flavour) through the unchanged
    driver.  The model is deterministic: same outer iteration count as the oracle."""
    n = 24
    out = _run(nranks, "--explicit_laplacian", "--set_1d_laplacian_size=%d" % n, "--num_iters=600",
               "--set_tol=1e-6", "--enable_onesided", "--enable_comm_overlap", "--global_convergence_type=decentralized")
    iters = sorted(set(int(x) for x in re.findall(r"converged in (\d+) iterations", out)))
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    ref = oracle.ras_run(rp, col, val, np.ones(N), nranks, oracle.first_rows_regular(N, nranks),
                         oracle.make_settings(max_iters=600, tol=1e-6, enable_onesided=1, enable_overlap=1))
    assert ref["converged"] and iters == [ref["iter_count"]], out
    rel = float(re.search(r"relative residual norm of solution ([0-9.eE+-]+)", out).group(1))
    assert abs(rel - ref["residual_norm"] / ref["rhs_norm"]) <= 1e-6 * rel + 1e-12


def test_bench_ras_reference_smoke_command(oracle):
    """The reference's own smoke run (schwarz.org:21): `mpirun -n 4 ./benchmarking/bench_ras
    --explicit_laplacian --enable_global_check`, every other flag at the driver's default
    (16 x 16 grid, 100 iterations cap, tolerance 1e-6, CG without preconditioner to 1e-12)."""
    out = _run(4, "--explicit_laplacian", "--enable_global_check")
    iters = sorted(set(int(x) for x in re.findall(r"converged in (\d+) iterations", out)))
    rp, col, val = oracle.laplacian2d(16)
    ref = oracle.ras_run(rp, col, val, np.ones(256), 4, oracle.first_rows_regular(256, 4),
                         oracle.make_settings(max_iters=100, tol=1e-6))
    if ref["converged"]:
        assert iters == [ref["iter_count"]], out
    else:
        assert "did not converge in 100 iterations" in out


def test_bench_ras_authors_iterative_study_settings(oracle):
    """The flag family of the authors' iterative-local-solve study (my_scripts/run_script:21-54):
    inexact local CG (local_tol 0.1, <= 70 iterations), block-Jacobi with the driver's default block
    size 16, overlap 8, one-sided with the decentralized convergence type, 100 outer iterations."""
    n, P = 48, 4
    out = _run(P, "--explicit_laplacian", "--set_1d_laplacian_size=%d" % n, "--num_iters=100", "--set_tol=1e-8",
               "--local_tol=0.1", "--local_max_iters=70", "--restart_iter=40", "--overlap=8",
               "--local_precond=block-jacobi", "--enable_onesided", "--global_convergence_type=decentralized",
               env={"SCHWZ_ONESIDED": "lockstep"})
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    ref = oracle.ras_run(rp, col, val, np.ones(N), P, oracle.first_rows_regular(N, P),
                         oracle.make_settings(max_iters=100, tol=1e-8, overlap=8, precond=2, precond_block_size=16,
                                              local_tol=0.1, local_max_iters=70, enable_onesided=1))
    if ref["converged"]:
        iters = sorted(set(int(x) for x in re.findall(r"converged in (\d+) iterations", out)))
        assert len(iters) == 1 and abs(iters[0] - ref["iter_count"]) <= 1, out
    else:
        assert out.count("did not converge in 100 iterations") == P, out


TYPES_BIN = os.path.join(ROOT, "schwarz-lib_amd", "build", "ras_types_driver")


def _run_types(nranks, types, n, mixed, overlapped, tol, max_iters, env=None, extra=()):
    """tests/drivers/ras_types_driver.cpp: the mirror's other template instantiations."""
    if not os.path.exists(TYPES_BIN):
        pytest.skip("ras_types_driver not built (`make -C schwarz-lib_amd types_driver`, needs MPI)")
    if not os.path.exists(MPIEXEC):
        pytest.skip("no mpiexec on this machine")
    cmd = [MPIEXEC, "-n", str(nranks), TYPES_BIN, types, str(n), str(int(mixed)), str(int(overlapped)),
           repr(tol), str(max_iters)] + list(extra)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stdout + p.stderr
    res = re.search(r"RESULT iters=(\d+) solnorm=([0-9.eE+-]+)", p.stdout)
    assert res, p.stdout
    return int(res.group(1)), float(res.group(2)), p.stdout


@pytest.mark.parametrize("overlapped", [False, True])
def test_mirror_mixed_precision_halos(oracle, overlapped):
    """SolverRAS<double, int32, float> with settings.use_mixed_precision: the packed halos cross
    the wire as fp32 (restricted_schwarz.cpp:898-903, 929-933, 952-954), over MPI staging here
    (ranks share the GPU).  The oracle rounds the same values the same way."""
    n, P = 24, 3
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    extra = dict(enable_onesided=1, enable_overlap=1) if overlapped else {}
    ref = oracle.ras_run(rp, col, val, np.ones(N), P, oracle.first_rows_regular(N, P),
                         oracle.make_settings(max_iters=400, tol=1e-4, use_mixed_precision=1, **extra))
    plain = oracle.ras_run(rp, col, val, np.ones(N), P, oracle.first_rows_regular(N, P),
                           oracle.make_settings(max_iters=400, tol=1e-4, **extra))
    it, norm, out = _run_types(P, "d32f", n, True, overlapped, 1e-4, 400)
    assert ref["converged"] and it == ref["iter_count"], out
    ref_norm = float(np.linalg.norm(ref["solution"]))
    assert abs(norm - ref_norm) <= 1e-10 * ref_norm
    # the fp32 rounding is visible: the fp64-halo answer differs from both
    assert abs(float(np.linalg.norm(plain["solution"])) - norm) > 1e-9 * ref_norm
    # MixedValueType = float without the setting keeps fp64 halos (the reference's default)
    it64, norm64, _ = _run_types(P, "d32f", n, False, overlapped, 1e-4, 400)
    assert it64 == plain["iter_count"]
    assert abs(norm64 - float(np.linalg.norm(plain["solution"]))) <= 1e-10 * ref_norm


@pytest.mark.parametrize("types", ["d64d", "d64f"])
def test_mirror_int64_index_instantiations(oracle, types):
    """IndexType = int64 (settings.hpp:533-537): same iteration as the int32 instantiation."""
    n, P = 20, 2
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    mixed = types.endswith("f")
    ref = oracle.ras_run(rp, col, val, np.ones(N), P, oracle.first_rows_regular(N, P),
                         oracle.make_settings(max_iters=400, tol=1e-4, use_mixed_precision=int(mixed)))
    it, norm, out = _run_types(P, types, n, mixed, False, 1e-4, 400)
    assert ref["converged"] and it == ref["iter_count"], out
    ref_norm = float(np.linalg.norm(ref["solution"]))
    assert abs(norm - ref_norm) <= 1e-10 * ref_norm


def test_bench_ras_onesided_rma_flavour_flags(oracle):
    """The reference's MPI RMA flavours (put instead of get, one message per value, local flush,
    local lock: bench_ras.cpp:73-97) and its default convergence type (centralized-tree) through the
    deterministic stand-in of the one-sided mode (SCHWZ_ONESIDED=lockstep: local tests, all-gathered
    flags), which the oracle can reproduce: the flags select calls, not values.  The free-running
    mode the same flags select by default is the next test."""
    n, P = 32, 3
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    ref = oracle.ras_run(rp, col, val, np.ones(N), P, oracle.first_rows_regular(N, P),
                         oracle.make_settings(max_iters=600, tol=1e-6, enable_onesided=1))
    assert ref["converged"]
    base = ["--explicit_laplacian", "--set_1d_laplacian_size=%d" % n, "--num_iters=600", "--set_tol=1e-6",
            "--enable_onesided"]
    for extra in ([], ["--remote_comm_type=put", "--enable_one_by_one", "--flush_type=flush-local",
                       "--lock_type=lock-local"]):
        out = _run(P, *(base + extra), env={"SCHWZ_ONESIDED": "lockstep"})
        iters = sorted(set(int(x) for x in re.findall(r"converged in (\d+) iterations", out)))
        assert iters == [ref["iter_count"]], out
        rel = float(re.search(r"relative residual norm of solution ([0-9.eE+-]+)", out).group(1))
        assert abs(rel - ref["residual_norm"] / ref["rhs_norm"]) <= 1e-6 * rel + 1e-12


@pytest.mark.parametrize("flavour", [("put", "centralized-tree", []), ("get", "decentralized", []),
                                     ("put", "decentralized", ["--enable_decentralized_accumulate"]),
                                     ("get", "centralized-tree", ["--enable_put_all_local_residual_norms=false"])])
def test_bench_ras_free_running_onesided(oracle, flavour):
    """--enable_onesided through the UNCHANGED reference driver, free running: three ranks on this GPU,
    halo values put into / got from the neighbours' device windows (HIP IPC), termination by the tree,
    the decentralised flags or the accumulated counters on an MPI shared-memory window.  Not
    deterministic: every rank converges (possibly at different iteration counts), and the residual
    of the assembled solution is what the tolerance implies."""
    n, P = 32, 3
    comm_type, conv_type, extra = flavour
    out = _run(P, "--explicit_laplacian", "--set_1d_laplacian_size=%d" % n, "--num_iters=3000", "--set_tol=1e-7",
               "--enable_onesided", "--remote_comm_type=%s" % comm_type, "--global_convergence_type=%s" % conv_type, *extra)
    assert "One-sided exchange: free running" in out, out
    iters = [int(x) for x in re.findall(r"converged in (\d+) iterations", out)]
    assert len(iters) == P and all(5 < i < 3000 for i in iters), out
    assert "did not converge" not in out
    rel = float(re.search(r"relative residual norm of solution ([0-9.eE+-]+)", out).group(1))
    assert rel < 1e-4
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    ref = oracle.ras_run(rp, col, val, np.ones(N), P, oracle.first_rows_regular(N, P),
                         oracle.make_settings(max_iters=3000, tol=1e-7))
    # asynchronous iterations need at least about as many sweeps as the synchronous loop
    assert ref["converged"] and max(iters) >= ref["iter_count"] - 2


@pytest.mark.parametrize("types", ["d32d", "d64d"])
def test_mirror_initialize_from_csr_arrays(oracle, types):
    """initialize(num_rows, row_ptrs, col_idxs, values, rhs): the caller's CSR system and right-hand
    side (the deal.II-free analogue of the reference's initialize(dealii::SparseMatrix, Vector),
    include/schwarz_base.hpp:96-97), for both index types, against the oracle on the same system."""
    n, P = 24, 3
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    rhs = 1.0 + (np.arange(N) % 7)
    ref = oracle.ras_run(rp, col, val, rhs, P, oracle.first_rows_regular(N, P),
                         oracle.make_settings(max_iters=500, tol=1e-8))
    it, norm, out = _run_types(P, types, n, False, False, 1e-8, 500, extra=("csr",))
    assert "Matrix handed over by the caller" in out
    assert ref["converged"] and it == ref["iter_count"], out
    ref_norm = float(np.linalg.norm(ref["solution"]))
    assert abs(norm - ref_norm) <= 1e-10 * ref_norm
    rel = float(re.search(r"relative residual norm of solution ([0-9.eE+-]+)", out).group(1))
    assert abs(rel - ref["residual_norm"] / ref["rhs_norm"]) <= 1e-6 * rel + 1e-12


@pytest.mark.parametrize("flags", [("--explicit_laplacian", "--set_1d_laplacian_size=48"),
                                   ("--matrix_filename=poisson3d:20x18x24", "--local_max_iters=6", "--local_tol=0")])
def test_bench_ras_early_exchange_is_bit_identical(tmp_path, flags):
    """The mirror's synchronous loop posts the exchange of the next iteration beside the tail of the local solve
    (schwz_ras_pack_early on a side stream; here the host-staged MPI transport, ranks sharing the GPU) --
    SCHWZ_EARLY_EXCHANGE=0 restores the reference's order.  The per-rank residual histories the unchanged
    driver writes are the same, digit for digit."""
    hist = {}
    for early in ("1", "0"):
        d = tmp_path / ("early" + early)
        d.mkdir()
        out = _run(3, *flags, "--enable_global_check", "--num_iters=300", "--set_tol=1e-7",
                   "--write_iters_and_residuals", cwd=str(d), env=dict(SCHWZ_EARLY_EXCHANGE=early))
        iters = set(int(x) for x in re.findall(r"converged in (\d+) iterations", out))
        assert len(iters) == 1 and min(iters) > 3, out
        rows = []
        for r in range(3):
            lines = (d / ("iter_res_%02d.csv" % r)).read_text().splitlines()[1:]
            rows.append([ln.split(",")[:2] for ln in lines])  # iteration, local residual norm (not the timestamp)
        hist[early] = (iters, rows, re.search(r"relative residual norm of solution ([0-9.eE+-]+)", out).group(1))
    assert hist["1"] == hist["0"]


@pytest.mark.parametrize("types", ["d32d", "d64d"])
def test_mirror_public_data_members(types):
    """The public data members of the reference's solver class (include/schwarz_base.hpp:137-197) on the
    mirror, checked by every rank of a direct-solver run (tests/drivers/ras_types_driver.cpp "members"):
    local_matrix / interface_matrix as host CSR, the factors with A(perm, perm) = L L^T and U = L^T,
    local_perm / local_inv_perm inverse to each other, local_solution = the rank's piece of the solution,
    the residual histories, global_matrix / global_rhs deliberately null."""
    n, P = 20, 3
    it, norm, out = _run_types(P, types, n, False, False, 1e-8, 300, extra=("members",))
    rows = re.findall(r"MEMBERS rank=(\d+) local_n=(\d+) local_size_x=(\d+) local_nnz=(\d+) iface_nnz=(\d+) iface_ok=(\d) "
                      r"factor_err=([0-9.eE+-]+) perm_ok=(\d) ut_ok=(\d) sol_err=([0-9.eE+-]+) global_null=(\d) "
                      r"global_solution=(\d) hist=(\d+)/(\d+) rhs0=([0-9.eE+-]+)", out)
    assert len(rows) == P, out
    for r in rows:
        rank, local_n, lsx, nnz, inz = (int(t) for t in r[:5])
        assert local_n == lsx and nnz > 4 * local_n and inz > 0 and r[5] == "1"
        assert float(r[6]) < 1e-12 and r[7] == "1" and r[8] == "1"
        assert float(r[9]) < 1e-12 * max(norm, 1.0)
        assert r[10] == "1" and int(r[11]) == (1 if rank == 0 else 0)
        assert int(r[12]) >= it and int(r[13]) == P and float(r[14]) == 1.0


def test_mirror_and_python_host_run_the_same_cg_launch_structure(schwz, torch_cuda):
    """The two hosts over the one C ABI must drive the same kernels: schwz_ras_cg_flavour (stored q / q-free,
    fused direction launch, deferred x, which launches walk) reported by the unchanged bench_ras on libschwz.so
    and by the Python host for the same problem and settings -- at a size where the walk applies."""
    import json
    sys_path = os.path.join(ROOT, "schwarz-lib_amd")
    flags = ["--matrix_filename=poisson3d:128", "--enable_global_check", "--set_tol=1e-30", "--num_iters=6",
             "--local_precond=block-jacobi", "--precond_max_block_size=1", "--local_max_iters=10", "--local_tol=0"]
    out = _run(1, *flags)
    m = re.search(r"outer loop: (\d+) iterations in ([0-9.eE+-]+) s, cg flavour (\d+)", out)
    assert m and int(m.group(1)) == 6, out
    s = schwz.Settings(laplacian_dim=3, laplacian_shape=(128, 128, 128))
    s.convergence_settings.enable_global_check = True
    md = schwz.Metadata(tolerance=1e-30, max_iters=6, local_precond="block-jacobi", precond_max_block_size=1,
                        local_solver_tolerance=0.0, local_max_iters=10, num_subdomains=1)
    solver = schwz.SolverRAS(s, md, comm=schwz.InProcessComm(1), quiet=True)
    solver.initialize()
    solver.run(gather_solution=False)
    flav = solver.subdomains[0].cg_flavour()
    assert flav == int(m.group(3)) and flav != 0, (flav, m.group(3))


def test_bench_ras_debug_dumps(oracle, tmp_path):
    """--print_matrices / --write_perm_data through the unchanged driver (schwarz_base.cpp:252-257, solve.cpp:401-450,
    utils.cpp:94-108): local_mat_<rank>.csv / int_mat_<rank>.csv as 1-based "row,col,value" lines, L_mat / U_mat and
    the factor permutation files of the direct local solver."""
    n, P = 12, 2
    _run(P, "--explicit_laplacian", "--set_1d_laplacian_size=%d" % n, "--enable_global_check", "--num_iters=200",
         "--set_tol=1e-7", "--local_solver=direct-ginkgo", "--print_matrices", "--write_perm_data", cwd=str(tmp_path))
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    for rank in range(P):
        osd = oracle.Subdomain(rp, col, val, P, rank, 2, oracle.first_rows_regular(N, P).astype(np.int32))
        lrp, lcol, lval = osd.local_matrix()
        rows = [l.split(",") for l in (tmp_path / ("local_mat_%d.csv" % rank)).read_text().splitlines()]
        assert len(rows) == len(lcol)
        got_r = np.array([int(r[0]) for r in rows]) - 1
        got_c = np.array([int(r[1]) for r in rows]) - 1
        got_v = np.array([float(r[2]) for r in rows])
        assert np.array_equal(got_r, np.repeat(np.arange(len(lrp) - 1), np.diff(lrp)))
        assert np.array_equal(got_c, lcol) and np.allclose(got_v, lval)
        assert (tmp_path / ("int_mat_%d.csv" % rank)).read_text().count("\n") == osd.nnz_interface
        perm = np.array([int(t) for t in (tmp_path / ("perm_%d.csv" % rank)).read_text().split()])
        inv = np.array([int(t) for t in (tmp_path / ("inv_perm_%d.csv" % rank)).read_text().split()])
        assert sorted(perm) == list(range(len(lrp) - 1)) and np.array_equal(inv[perm], np.arange(len(perm)))
        L = (tmp_path / ("L_mat_%d.csv" % rank)).read_text().splitlines()
        U = (tmp_path / ("U_mat_%d.csv" % rank)).read_text().splitlines()
        assert len(L) == len(U) >= len(lrp) - 1
        assert all(int(l.split(",")[0]) >= int(l.split(",")[1]) for l in L)   # lower triangular
