#!/usr/bin/env python3
"""Host time per outer iteration of the N = 1 bench workload: how long each call of SolverRAS.step() keeps the
host busy (the launches of a whole local solve are enqueued by ONE C call), and how long the host waits for
the 8-byte norm.  A host that needs more than the GPU's ~1.85 ms per step becomes the bottleneck."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))
import torch
import schwz_amd as S
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
solver, m = bench.make_solver(S, S.InProcessComm(1), (n, n, n), 10, 1e-30, 200, 0.0, 0)
sd = solver.subdomains[0]
stream = solver.backend.stream()
solver.begin_run()
for _ in range(5):
    solver.step()
torch.cuda.synchronize()
acc = dict(update_boundary=0.0, launch=0.0, wait=0.0, restrict=0.0, step=0.0)
K = 50
t_all = time.perf_counter()
for _ in range(K):
    t0 = time.perf_counter()
    sd.update_boundary(stream)
    t1 = time.perf_counter()
    sd.check_and_solve_launch(stream)
    t2 = time.perf_counter()
    sd.local_residual_wait()
    t3 = time.perf_counter()
    sd.restrict(stream)
    t4 = time.perf_counter()
    acc["update_boundary"] += t1 - t0
    acc["launch"] += t2 - t1
    acc["wait"] += t3 - t2
    acc["restrict"] += t4 - t3
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print("raw C calls : %.3f ms per step; host busy: boundary %.1f us, solve launch %.1f us, restrict %.1f us; waiting for the norm %.1f us"
      % (1e3 * tot / K, 1e6 * acc["update_boundary"] / K, 1e6 * acc["launch"] / K, 1e6 * acc["restrict"] / K, 1e6 * acc["wait"] / K))
solver.begin_run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    solver.step()
torch.cuda.synchronize()
print("SolverRAS.step(): %.3f ms per step" % (1e3 * (time.perf_counter() - t0) / K))
