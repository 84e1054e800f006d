"""Edge cases of the host layer the advisor listed after round 1: a zero right-hand side must
not raise (the reference gets NaN from 0/0 and keeps iterating), the overlapped mode refuses more than 62
subdomains like the C++ mirror, a custom partition needs a valid partition vector.  CPU only: the
per-subdomain arithmetic is the test-only oracle backend."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _solver(schwz, P, settings=None, **md):
    from oracle_backend import OracleBackend
    s = settings or schwz.Settings()
    m = schwz.Metadata(num_subdomains=P, **md)
    return schwz.SolverRAS(s, m, comm=schwz.InProcessComm(P), backend=OracleBackend(), quiet=True), m


def test_zero_right_hand_side_iterates_without_raising(schwz, oracle):
    solver, m = _solver(schwz, 2, oned_laplacian_size=8, tolerance=1e-6, max_iters=5)
    solver.initialize(matrix=oracle.laplacian2d(8), rhs=np.zeros(64))
    out = solver.run()
    # 0 / 0 = NaN never passes "<= tol": all iterations are done, like the reference and the mirror
    assert out["iter_count"] == 5 and not out["converged"]
    assert np.all(out["solution"] == 0.0)


def test_ratio_follows_ieee_division():
    from schwz_amd.solver import _ratio
    assert np.isnan(_ratio(0.0, 0.0)) and _ratio(1.0, 0.0) == float("inf") and _ratio(1.0, 4.0) == 0.25
    assert np.isnan(_ratio(float("nan"), 0.0))


def test_overlapped_mode_refuses_more_than_62_subdomains(schwz):
    s = schwz.Settings()
    s.comm_settings.enable_onesided = True
    s.comm_settings.enable_overlap = True
    s.convergence_settings.enable_decentralized_leader_election = True
    solver, m = _solver(schwz, 63, settings=s, oned_laplacian_size=16, tolerance=1e-6, max_iters=3)
    solver.initialize()
    with pytest.raises(schwz.NotImplementedSchwz):
        solver.run()


def test_custom_partition_needs_a_valid_vector(schwz):
    for vec in (None, np.zeros(5, dtype=np.uint32), np.full(64, 7, dtype=np.uint32)):
        s = schwz.Settings(partition=schwz.PARTITION_CUSTOM, partition_vector=vec)
        solver, m = _solver(schwz, 2, settings=s, oned_laplacian_size=8)
        with pytest.raises(schwz.SchwzError):
            solver.initialize()


def test_onesided_run_says_which_protocol_runs(schwz, capsys):
    from oracle_backend import OracleBackend
    s = schwz.Settings()
    s.comm_settings.enable_onesided = True
    s.comm_settings.enable_put = True
    s.convergence_settings.enable_global_simple_tree = True
    m = schwz.Metadata(num_subdomains=2, oned_laplacian_size=8, tolerance=1e-6, max_iters=50)
    solver = schwz.SolverRAS(s, m, comm=schwz.InProcessComm(2), backend=OracleBackend(), quiet=False)
    solver.initialize()
    solver.run()
    out = capsys.readouterr().out
    assert "RMA flavour flags" in out and "without node windows" in out


def test_partition_debug_file(tmp_path, monkeypatch):
    """write_debug_out (partition_tools.hpp:96-106): part_indices.csv of a permuting partition, written before any
    device is touched."""
    import numpy as np
    import schwz_amd as S
    monkeypatch.chdir(tmp_path)
    s = S.Settings(partition=S.PARTITION_REGULAR2D, write_debug_out=True)
    m = S.Metadata(oned_laplacian_size=8, num_subdomains=4)

    class _NoGpu:
        problem_laplacian = staticmethod(S.Problem.laplacian)
        partition_regular = staticmethod(S.partition_regular)
        partition_regular2d = staticmethod(S.partition_regular2d)

    solver = S.SolverRAS(s, m, comm=S.InProcessComm(4), backend=_NoGpu(), quiet=True)
    prob = solver._partition(solver._setup_global_matrix())
    lines = (tmp_path / "part_indices.csv").read_text().splitlines()
    assert lines[0] == "idx,subd" and len(lines) == 65
    part = np.array([int(l.split(",")[1]) for l in lines[1:]])
    assert np.array_equal(np.bincount(part), [16, 16, 16, 16]) and prob.N == 64


def test_setup_threads_follow_the_environment(tmp_path):
    """Threads of the host-side setup loops (schwz_setup_threads): the explicit setting wins; otherwise the CPUs of the
    process, capped by its cgroup quota and 32, divided among the ranks of a node a launcher announces.  Fixed at first
    use, so every case runs in a process of its own."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "schwarz-lib_amd")
    code = ("import sys; sys.path.insert(0, %r); import schwz_amd._capi as c; print(c.lib.schwz_setup_threads())" % root)

    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("SCHWZ_SETUP_THREADS", "LOCAL_WORLD_SIZE",
                                                              "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS")}
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=120)
        assert p.returncode == 0, p.stderr
        return int(p.stdout.strip().splitlines()[-1])

    base = run()
    assert 1 <= base <= 32 and base <= len(os.sched_getaffinity(0))
    assert run(SCHWZ_SETUP_THREADS="5") == 5
    assert run(LOCAL_WORLD_SIZE="2") == max(1, base // 2)
    assert run(LOCAL_WORLD_SIZE="64") == max(1, base // 64)
    assert run(SCHWZ_SETUP_THREADS="3", LOCAL_WORLD_SIZE="8") == 3
