#!/usr/bin/env python3
"""Cost of the halo steps (pack, exchange, unpack, boundary update) next to the local solve: P z-slabs
of 512 x 512 x 64 in ONE process on one GPU (exchange = device copies), per-subdomain step time for
P = 1, 2, 4 and the time_struct shares the host sees."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))
import torch  # noqa: E402
import schwz_amd as S  # noqa: E402
import bench  # noqa: E402

for P in ([int(t) for t in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 2, 4)):
    comm = S.InProcessComm(P)
    solver, m = bench.make_solver(S, comm, (512, 512, 64 * P), 10, 1e-30, 100, 0.0, 0,
                                  overlapped=os.environ.get("HALO_PROBE_OVERLAPPED") == "1")
    solver.begin_run()
    for _ in range(3):
        solver.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        solver.step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / K
    print("P=%d: %.3f ms per outer iteration, %.3f ms per subdomain" % (P, 1e3 * el, 1e3 * el / P), flush=True)
    del solver
    torch.cuda.empty_cache()
