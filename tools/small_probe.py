#!/usr/bin/env python3
"""Launch-bound regime: BASELINE configs[0] (2-D Poisson 256^2, 2 subdomains, converged local CG)
on one GPU -- time per outer iteration and per inner CG iteration."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))
import torch  # noqa: E402
import schwz_amd as S  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
s = S.Settings()
s.convergence_settings.enable_global_check = True
m = S.Metadata(num_subdomains=P, oned_laplacian_size=256, tolerance=1e-6, max_iters=500,
               local_precond="block-jacobi", precond_max_block_size=1)
solver = S.SolverRAS(s, m, comm=S.InProcessComm(P), quiet=True)
solver.initialize()
torch.cuda.synchronize()
t0 = time.perf_counter()
out = solver.run(gather_solution=False)
torch.cuda.synchronize()
el = time.perf_counter() - t0
sd = solver.subdomains[0]
# inner iterations of one representative local solve
import ctypes
it = ctypes.c_int(0)
S.capi.check(S.capi.lib.schwz_ras_local_solve(sd.h, ctypes.byref(it), 0))
print("outer iterations %d, converged %s, %.3f s total, %.2f ms per outer iteration" %
      (out["iter_count"], out["converged"], el, 1e3 * el / max(out["iter_count"], 1)))
print("a converged local solve: %d CG iterations" % it.value)
