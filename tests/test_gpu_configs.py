"""BASELINE.json's configurations at their own sizes on the GPU.

configs[0] (2-D 256^2, 2 subdomains, two-sided, reference-default local solve) is small enough
for the oracle to be the checker at full size.  configs[4] (1024^3 in 8 slabs, one-sided
overlapped exchange, decentralised stop) is checked at the per-GPU size of that configuration
(two slabs of 1024 x 1024 x 128 held by this one GPU) through size-independent properties, and
against the oracle on a small 3-D problem of the same shape.  configs[1]/[2] live in
test_gpu_ras.py::test_baseline_full_size_properties, configs[3] in
test_ras_ani4_direct_eight_subdomains."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _overlapped_settings(schwz, **kw):
    s = schwz.Settings(**kw)
    s.comm_settings.enable_onesided = True
    s.comm_settings.enable_overlap = True
    s.convergence_settings.enable_decentralized_leader_election = True
    return s


def _oracle_overlapped(oracle, csr, P, first_row, **kw):
    rp, col, val = csr
    N = len(rp) - 1
    st = oracle.make_settings(enable_onesided=1, enable_overlap=1, enable_global_check=1, **kw)
    return oracle.ras_run(rp, col, val, np.ones(N), P, np.asarray(first_row, dtype=np.int32), st)


def test_config0_full_size_matches_oracle(schwz, oracle, torch_cuda):
    """BASELINE configs[0] at its own size: 2-D Poisson 256 x 256, regular-1D partition, 2
    subdomains, iterative local solve at the reference defaults (local_tol 1e-12, unlimited
    inner iterations, no preconditioner), two-sided exchange, --set_tol=1e-6
    --enable_global_check.  Same outer iteration count as the oracle (+-1: both stop where the
    criterion crosses 1e-6 and the local solves differ at rounding level), residual history and
    solution to fp64 tolerance, structural known answers of SURVEY Appendix B (C1)."""
    n, P = 256, 2
    s = schwz.Settings()
    m = schwz.Metadata(num_subdomains=P, oned_laplacian_size=n, tolerance=1e-6, max_iters=2000)
    solver = schwz.SolverRAS(s, m, comm=schwz.InProcessComm(P), quiet=True)
    solver.initialize()
    sd0 = solver.subdomains[0]
    assert (sd0.local_size, sd0.overlap_size, sd0.local_size_x, sd0.nnz_interface) == (32768, 256, 33024, 256)
    assert sd0.nnz_local == 164350 and sd0.num_recv == 512
    out = solver.run()
    rp, col, val = oracle.laplacian2d(n)
    N = n * n
    import os
    st = oracle.make_settings(max_iters=2000, tol=1e-6, num_threads=min(16, os.cpu_count() or 1))
    r = oracle.ras_run(rp, col, val, np.ones(N), P, oracle.first_rows_regular(N, P), st)
    assert out["converged"] and r["converged"]
    assert abs(out["iter_count"] - r["iter_count"]) <= 1
    hist = np.array(m.post_process_data["global_residual_vector_out"]).sum(axis=0)
    k = min(len(hist), len(r["hist_global"]))
    assert k >= 300
    assert np.abs(hist[:k] - r["hist_global"][:k]).max() <= 1e-9 * r["hist_global"][0]
    assert np.abs(out["solution"] - r["solution"]).max() <= 1e-8 * np.abs(r["solution"]).max()
    assert out["residual_norm"] / out["rhs_norm"] < 1e-5
    assert abs(out["residual_norm"] - r["residual_norm"]) <= 1e-6 * r["rhs_norm"]


@pytest.mark.parametrize("P", [2, 4])
@pytest.mark.parametrize("local", ["truncated", "converged"])
def test_overlapped_3d_slabs_match_oracle(schwz, oracle, torch_cuda, P, local):
    """The configs[4] flavour on a small 3-D slab partition against the oracle: halos posted before
    the local solve and consumed one iteration later, local tests, flooded (mask, stop) flags.
    `truncated` = the bench operating point (10 CG iterations + Jacobi, local_tol 0), `converged`
    = local solves to 1e-10.  Iteration for iteration: same stop, same residual history."""
    shape = (18, 14, 32)
    trunc = local == "truncated"
    s = _overlapped_settings(schwz, laplacian_dim=3, laplacian_shape=shape)
    m = schwz.Metadata(num_subdomains=P, tolerance=1e-5, max_iters=900, local_precond="block-jacobi",
                       precond_max_block_size=1, local_solver_tolerance=0.0 if trunc else 1e-10,
                       local_max_iters=10 if trunc else -1)
    solver = schwz.SolverRAS(s, m, comm=schwz.InProcessComm(P), quiet=True)
    solver.initialize()
    out = solver.run()
    r = _oracle_overlapped(oracle, oracle.laplacian3d(*shape), P, m.first_row, max_iters=900, tol=1e-5,
                           precond=oracle.PRECOND_JACOBI, local_tol=0.0 if trunc else 1e-10,
                           local_max_iters=10 if trunc else -1)
    assert out["converged"] and r["converged"]
    if trunc:
        # fixed-work local solves make the outer map rounding sensitive: first
        # iterations tight, the stop within two iterations
        assert abs(out["iter_count"] - r["iter_count"]) <= 2
    else:
        assert out["iter_count"] == r["iter_count"]
    loc = np.array(m.post_process_data["local_residual_vector_out"]).reshape(-1, P)
    ref = np.asarray(r["hist_local"])
    k = min(len(loc), len(ref), 6 if trunc else 10 ** 9)
    assert k >= 6 and np.abs(loc[:k] - ref[:k]).max() <= 1e-9 * ref[0].max()
    scale = np.abs(r["solution"]).max()
    assert np.abs(out["solution"] - r["solution"]).max() <= (1e-3 if trunc else 1e-7) * scale
    assert out["residual_norm"] / out["rhs_norm"] < 1e-3


def _stencil_residual_norm(torch, x_host, shape):
    """|| 1 - A x ||_2 of the 7-point Dirichlet Laplacian (diag 6, off -1, x fastest) with plain
    torch slicing: shares no code with the library."""
    nx, ny, nz = shape
    X = torch.from_numpy(x_host).cuda().view(nz, ny, nx)
    R = 1.0 - 6.0 * X
    R[:, :, 1:] += X[:, :, :-1]
    R[:, :, :-1] += X[:, :, 1:]
    R[:, 1:, :] += X[:, :-1, :]
    R[:, :-1, :] += X[:, 1:, :]
    R[1:, :, :] += X[:-1, :, :]
    R[:-1, :, :] += X[1:, :, :]
    return float(torch.linalg.norm(R.view(-1)))


def test_config4_per_gpu_size_overlapped_properties(schwz, oracle, torch_cuda):
    """BASELINE configs[4] (1024^3, 8 subdomains, asynchronous RAS with decentralised convergence,
    halo overlapped with the local solve) at ITS per-GPU size: two z-slabs of 1024 x 1024 x 128
    (134 M interior rows each, + overlap planes) held by this one GPU, one-sided overlapped mode,
    10 CG iterations + Jacobi per local solve.  The oracle cannot run this size; checked instead:
    sizes of SURVEY Appendix B (C5 end slab); the run does not stop early and every subdomain
    records every iteration; the reported true residual equals an independent recomputation (plain
    torch stencil on the assembled solution); the solution is finite, bounded by the discrete
    maximum principle's bound and symmetric under reversal of the ordering (the two slabs mirror
    each other); and the stop agreement of the flooding protocol fires at the iteration the oracle
    predicts for the same number of subdomains."""
    torch = torch_cuda
    free, _ = torch.cuda.mem_get_info()
    if free < 90e9:
        pytest.skip("needs ~75 GB of HBM")
    shape, P, K = (1024, 1024, 256), 2, 8
    s = _overlapped_settings(schwz, laplacian_dim=3, laplacian_shape=shape)
    m = schwz.Metadata(num_subdomains=P, tolerance=1e-6, max_iters=K, local_precond="block-jacobi",
                       precond_max_block_size=1, local_solver_tolerance=0.0, local_max_iters=10)
    solver = schwz.SolverRAS(s, m, comm=schwz.InProcessComm(P), quiet=True)
    solver.initialize()
    sd0, sd1 = solver.subdomains[0], solver.subdomains[1]
    plane = shape[0] * shape[1]
    for sd in (sd0, sd1):  # SURVEY Appendix B, "C5 3-D 1024^3 slab, end"
        assert (sd.local_size, sd.overlap_size, sd.halo_size, sd.local_size_x) == \
            (134217728, plane, plane, 135266304)
        assert sd.nnz_local == 944238592 and sd.nnz_interface == plane and sd.num_recv == 2 * plane
    out = solver.run()
    assert not out["converged"] and out["iter_count"] == K
    loc = np.array(m.post_process_data["local_residual_vector_out"]).reshape(-1, P)
    assert loc.shape == (K, P) and np.isfinite(loc).all() and (loc > 0).all()
    # mirrored slabs see mirrored problems: equal local residual norms (summation order aside)
    assert np.abs(loc[:, 0] - loc[:, 1]).max() <= 1e-6 * loc.max()
    x = out["solution"]
    N = shape[0] * shape[1] * shape[2]
    assert x.size == N
    res = _stencil_residual_norm(torch, x, shape)
    assert abs(res - out["residual_norm"]) <= 1e-8 * out["rhs_norm"]
    assert abs(out["rhs_norm"] - np.sqrt(N)) <= 1e-9 * np.sqrt(N)
    # (eight outer iterations of ten CG steps are far from converged at this size: the iterate still
    # carries a sawtooth of ~2e-4 of the solution scale next to the boundary -- the oracle shows the
    # same relative amplitude at 128 x 128 x 32 -- so positivity is NOT a property to assert here)
    assert np.isfinite(x).all() and np.abs(x).max() <= 3.0 * (shape[0] + 1) ** 2 / 8.0
    assert np.abs(x - x[::-1]).max() <= 1e-6 * np.abs(x).max()
    del x
    torch.cuda.empty_cache()
    # stop agreement at full size: a tolerance every subdomain meets at its first test makes the
    # flooded masks fill as fast as the neighbour graph allows; all subdomains must leave the loop
    # at the one iteration the protocol agrees on -- the iteration the oracle reaches on a small
    # grid with the same number of subdomains (the protocol does not depend on the size)
    m.tolerance, m.max_iters = 1e30, 20
    out2 = solver.run(gather_solution=False)
    small = (12, 10, 16)
    r = _oracle_overlapped(oracle, oracle.laplacian3d(*small), P,
                           oracle.first_rows_regular(small[0] * small[1] * small[2], P), max_iters=20, tol=1e30,
                           precond=oracle.PRECOND_JACOBI, local_tol=0.0, local_max_iters=10)
    assert out2["converged"] and r["converged"]
    assert out2["iter_count"] == r["iter_count"] == 3
    assert solver._stop[0] == solver._stop[1] == 3


@pytest.mark.parametrize("case", ["cube_256", "middle_slab_512x512x64"])
def test_z_sweep_walk_at_bench_size_is_bit_identical_per_row(schwz, torch_cuda, monkeypatch, case):
    """The z-sweep walk of the CG launches at the sizes the bench runs: the 256^3 matrix of configs[1]
    and the local matrix of a MIDDLE slab of the 512 x 512 x 64N grids (two appended overlap planes,
    chained to the interior: nothing left to a companion launch).  One CG iteration -- every row's
    q_i = (A p)_i, the r update and x += alpha p -- gives the same bits with the walk and with the
    chunk-by-chunk gather launches; ten iterations agree to the order in which partial sums are
    folded.  A middle slab's solve inside the RAS loop starts with the fused dual residual, a stand-alone
    solve (this test) in the walk (flavour bit 32)."""
    torch = torch_cuda
    if case == "cube_256":
        prob = schwz.Problem.laplacian(3, 256, 256, 256)
        sd = schwz.Subdomain(prob, 1, 0, 2, schwz.partition_regular(prob.N, 1))
    else:
        prob = schwz.Problem.laplacian(3, 512, 512, 192)
        sd = schwz.Subdomain(prob, 3, 1, 2, schwz.partition_regular(prob.N, 3))
    rp, col, val = sd.local_matrix()
    n = len(rp) - 1
    A = schwz.Csr(rp, col, val)
    del rp, col, val
    assert A.format() == 3 and A.sweep_slots() > 0 and A.sweep_left_out() == 0
    cg = schwz.Pcg(A, 1)
    g = torch.Generator(device="cuda").manual_seed(5)
    b = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    x0 = 0.1 * torch.randn(n, dtype=torch.float64, device="cuda", generator=g)

    def solve(sweep, iters, start="0"):
        monkeypatch.setenv("SCHWZ_CG_SWEEP", sweep)
        monkeypatch.setenv("SCHWZ_CG_SWEEPSTART", start)
        x = x0.clone()
        it, rn = cg.solve(b.data_ptr(), x.data_ptr(), 0.0, iters)
        return rn, x

    rn0, x_ref = solve("0", 1)
    assert cg.flavour() & 24 == 0
    rn1, x_sw = solve("1", 1)
    assert cg.flavour() & 56 == 24
    assert torch.equal(x_ref, x_sw) and abs(rn0 - rn1) <= 1e-13 * rn0
    # the solve STARTED in the walk as well (the default): rho_0 is folded from other partial sums, so one
    # iteration agrees to rounding instead of bit for bit
    rn2, x_st = solve("1", 1, start="1")
    assert cg.flavour() & 56 == 56
    assert float((x_ref - x_st).abs().max()) <= 1e-13 * float(x_ref.abs().max()) and abs(rn0 - rn2) <= 1e-13 * rn0
    rn0, x_ref = solve("0", 10)
    for start in ("0", "1"):
        rn1, x_sw = solve("1", 10, start=start)
        assert float((x_ref - x_sw).abs().max()) <= 1e-12 * float(x_ref.abs().max())
        assert abs(rn0 - rn1) <= 1e-10 * rn0


_FREE_RUNNING_AT_SIZE = r"""
import json, os, sys
sys.path.insert(0, %(pkg)r)
import os

import numpy as np
import torch, torch.distributed as dist
import schwz_amd as S
torch.cuda.set_device(0)
dist.init_process_group("gloo")
comm = S.WindowComm(device=torch.device("cuda", 0))
shape = %(shape)r
s = S.Settings(laplacian_dim=3, laplacian_shape=shape)
s.comm_settings.enable_onesided = True
s.comm_settings.enable_put = True
s.comm_settings.enable_get = False
s.convergence_settings.enable_decentralized_leader_election = True
m = S.Metadata(tolerance=1e-30, max_iters=%(iters)d, local_precond="block-jacobi", precond_max_block_size=1,
               local_solver_tolerance=0.0, local_max_iters=10)
solver = S.SolverRAS(s, m, comm=comm, quiet=True)
solver.initialize()
out = solver.run(gather_solution=False)
sd = solver.subdomains[comm.rank]
# this rank's part of || 1 - A x ||^2 with plain torch slicing on ITS x~ (interior planes + the overlap plane next to
# them, fresh from the exchange finish_run does): no kernel of the library touches the check, only a copy out of x~
nx, ny, nz = shape
plane = nx * ny
ptr, n = sd.vector(0)
flat = torch.empty(n, dtype=torch.float64, device="cuda")
idx = torch.arange(n, dtype=torch.int32, device="cuda")
S.gather(n, idx.data_ptr(), ptr, flat.data_ptr())
torch.cuda.synchronize()
del idx
ni = sd.local_size // plane
interior, overlap = flat[:sd.local_size].view(ni, ny, nx), flat[sd.local_size:sd.local_size + plane].view(1, ny, nx)
X = torch.cat((interior, overlap), 0) if comm.rank == 0 else torch.cat((overlap, interior), 0)
del flat
R = 1.0 - 6.0 * X
R[:, :, 1:] += X[:, :, :-1]
R[:, :, :-1] += X[:, :, 1:]
R[:, 1:, :] += X[:, :-1, :]
R[:, :-1, :] += X[:, 1:, :]
R[1:, :, :] += X[:-1, :, :]
R[:-1, :, :] += X[1:, :, :]
mine = R[:ni] if comm.rank == 0 else R[1:]
local_sq = float((mine * mine).sum())
loc = [float(v) for v in m.post_process_data["local_residual_vector_out"]]
print(json.dumps(dict(rank=comm.rank, iters=out["iter_count"], conv=bool(out["converged"]), local_sq=local_sq,
                      residual_norm=out["residual_norm"], rhs_norm=out["rhs_norm"], local_size=int(sd.local_size),
                      overlap=int(sd.overlap_size), halo=int(sd.halo_size), nnz=int(sd.nnz_local),
                      flavour=int(sd.cg_flavour()), loc=loc, finite=bool(np.isfinite(loc).all()))),
      flush=True)
solver.close()
dist.destroy_process_group()
"""


def test_config4_per_gpu_size_free_running_mode(torch_cuda, tmp_path):
    """BASELINE configs[4] at its per-GPU size in the FREE-RUNNING one-sided mode (the reference's asynchronous
    iteration, restricted_schwarz.cpp:715-852; round 2 ran this mode at 24 x 20 x 30 only): two rank PROCESSES
    share this GPU, each holds a 1024 x 1024 x 128 slab (134 M interior rows, 944 238 592 local nonzeros), the
    halos are pack kernels storing into the neighbour's HIP-IPC window, no collective in the loop.  The mode is
    non-deterministic by construction; checked by properties: both ranks leave the loop at the iteration cap and
    not before, on the walk kernels with the deferred x update (flavour bits), local residuals finite, the first one
    ||b_loc||, decaying after the jump of the first solves, and the true residual the library reports equals the one recomputed per rank with plain torch
    slicing on that rank's x~ (interior planes and the overlap plane received from the neighbour)."""
    import json
    import socket
    import subprocess
    import sys
    torch = torch_cuda
    free, _ = torch.cuda.mem_get_info()
    if free < 110e9:
        pytest.skip("needs ~90 GB of HBM")
    shape, iters, world = (1024, 1024, 256), 12, 2
    script = tmp_path / "rank.py"
    script.write_text(_FREE_RUNNING_AT_SIZE % dict(
        pkg=os.path.join(os.path.dirname(os.path.dirname(__file__)), "schwarz-lib_amd"), shape=shape, iters=iters))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            o, e = p.communicate()
        assert p.returncode == 0, o + e[-3000:]
        outs.append(json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1]))
    outs.sort(key=lambda d: d["rank"])
    plane = shape[0] * shape[1]
    for o in outs:
        assert not o["conv"] and o["iters"] == iters
        assert (o["local_size"], o["overlap"], o["halo"], o["nnz"]) == (134217728, plane, plane, 944238592)
        assert o["flavour"] & 28 == 28, o["flavour"]     # deferred x, both CG launches walk
        # (truncated local solves: the local residual jumps after the first solves and decays from there, like in
        # the synchronous run of test_baseline_full_size_properties)
        loc = o["loc"]
        assert o["finite"] and len(loc) == iters and min(loc) > 0.0
        assert abs(loc[0] - np.sqrt(o["local_size"] + o["overlap"])) <= 1e-9 * loc[0]   # x0 = 0: the first residual is ||b_loc||
        assert loc[-1] < max(loc[1:4])
    N = shape[0] * shape[1] * shape[2]
    assert abs(outs[0]["rhs_norm"] - np.sqrt(N)) <= 1e-9 * np.sqrt(N)
    res = np.sqrt(outs[0]["local_sq"] + outs[1]["local_sq"])
    assert abs(res - outs[0]["residual_norm"]) <= 1e-8 * outs[0]["rhs_norm"]
    assert outs[0]["residual_norm"] == outs[1]["residual_norm"]
