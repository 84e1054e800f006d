#!/usr/bin/env python3
"""Where the (untimed) setup of a run goes: SCHWZ_SETUP_TIMING=1 makes the library print the wall time of its
setup stages; this runs initialize() of the N = 1 workload (or one z-slab: --shape nx,ny,nz) and sums them up.

    python tools/setup_probe.py [--shape 256,256,256] [--subdomains 1]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SCHWZ_SETUP_TIMING"] = "1"
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256,256,256")
ap.add_argument("--subdomains", type=int, default=1)
a = ap.parse_args()
import torch
import schwz_amd as S
torch.cuda.init()
torch.zeros(1, device="cuda")
shape = tuple(int(t) for t in a.shape.split(","))
s = S.Settings(laplacian_dim=3, laplacian_shape=shape, overlap=2, partition=S.PARTITION_REGULAR)
s.convergence_settings.enable_global_check = True
m = S.Metadata(tolerance=1e-30, max_iters=10, local_precond="block-jacobi", precond_max_block_size=1,
               local_solver_tolerance=0.0, local_max_iters=10, num_subdomains=a.subdomains)
t0 = time.perf_counter()
solver = S.SolverRAS(s, m, comm=S.InProcessComm(a.subdomains), quiet=True)
solver.initialize()
torch.cuda.synchronize()
t1 = time.perf_counter()
solver.begin_run()
solver.step()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("initialize %.3f s, first step (first-use allocations) %.3f s, shape %s, %d subdomain(s)" %
      (t1 - t0, t2 - t1, shape, a.subdomains), flush=True)
