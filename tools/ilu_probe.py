#!/usr/bin/env python3
"""CG + ILU(0) on a 3-D Poisson matrix: time per iteration (the preconditioner is two level-scheduled
triangular sweeps, ~3 N^(1/3) launches each; SCHWZ_TRS_GRAPH=0|1 replays them launch by launch or
as one hipGraph).   python tools/ilu_probe.py [edge] [precond: 3 ilu | 4 isai | 1 jacobi]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))
import torch  # noqa: E402
import schwz_amd as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
pc = int(sys.argv[2]) if len(sys.argv) > 2 else 3
prob = S.Problem.laplacian(3, n, n, n)
sd = S.Subdomain(prob, 1, 0, 2, S.partition_regular(prob.N, 1))
rp, col, val = sd.local_matrix()
A = S.Csr(rp, col, val)
t0 = time.perf_counter()
cg = S.Pcg(A, pc)
setup = time.perf_counter() - t0
b = torch.ones(prob.N, dtype=torch.float64, device="cuda")
x = torch.zeros(prob.N, dtype=torch.float64, device="cuda")
cg.solve(b.data_ptr(), x.data_ptr(), 0.0, 3)
x.zero_()
cg.solve(b.data_ptr(), x.data_ptr(), 0.0, 20)   # same length as the timed solve: records its hipGraph, if any
torch.cuda.synchronize()
x.zero_()
t0 = time.perf_counter()
it, rn = cg.solve(b.data_ptr(), x.data_ptr(), 0.0, 20)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("SCHWZ_TRS_FLAGS=%s SCHWZ_TRS_FLAG_GRID=%s" % (os.environ.get("SCHWZ_TRS_FLAGS"), os.environ.get("SCHWZ_TRS_FLAG_GRID")))
print("n=%d^3 precond=%d setup %.2f s, %d iterations, %.3f ms per iteration, resnorm %.3e" %
      (n, pc, setup, it, 1e3 * el / it, rn))
