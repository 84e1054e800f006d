#!/usr/bin/env python3
"""Step time of the fixed-work RAS iteration on a large 2-D 5-point Laplacian (the reference's own generator,
initialization.cpp:214-265), one subdomain: z-sweep walk with the x line as the plane against the chunk-by-chunk
launches (SCHWZ_CG_SWEEP=0 is read per launch).

    python tools/walk2d_probe.py [n1d ...]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))
import torch
import schwz_amd as S
for n1d in [int(t) for t in sys.argv[1:]] or [4096]:
    s = S.Settings()
    s.convergence_settings.enable_global_check = True
    m = S.Metadata(oned_laplacian_size=n1d, tolerance=1e-30, max_iters=200, local_precond="block-jacobi",
                   precond_max_block_size=1, local_solver_tolerance=0.0, local_max_iters=10, num_subdomains=1)
    solver = S.SolverRAS(s, m, comm=S.InProcessComm(1), quiet=True)
    solver.initialize()
    sd = solver.subdomains[0]
    out = {}
    for mode in ("1", "0", "1", "0"):
        os.environ["SCHWZ_CG_SWEEP"] = mode
        solver.begin_run()
        for _ in range(3):
            solver.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            solver.step()
        torch.cuda.synchronize()
        out.setdefault(mode, []).append(round((time.perf_counter() - t0) / 20 * 1e3, 4))
        flav = sd.cg_flavour()
        out.setdefault("flavour" + mode, flav)
    print(n1d, "x", n1d, "rows", sd.local_size_x, "ms/step walk", out["1"], "chunk kernels", out["0"],
          "flavours", out["flavour1"], out["flavour0"], flush=True)
    del solver, sd
    torch.cuda.empty_cache()
