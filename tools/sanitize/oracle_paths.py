import os, sys, ctypes, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import oracle
oracle._SO = os.environ.get("SCHWZ_ASAN_ORACLE", "/tmp/libschwz_oracle_asan.so")
oracle._lib = None
rp, col, val = oracle.laplacian3d(10, 9, 8)
N = len(rp) - 1
for kw in (dict(precond=1), dict(precond=2, precond_block_size=7), dict(precond=3), dict(precond=4),
           dict(local_solver=oracle.SOLVER_DIRECT), dict(enable_onesided=1, enable_overlap=1),
           dict(non_symmetric=1, restart_iter=5, precond=1), dict(use_mixed_precision=1),
           dict(precond=1, local_tol=0.0, local_max_iters=3, reset_local_crit_iter=2, updated_max_iters=9)):
    for P in (1, 3):
        r = oracle.ras_run(rp, col, val, np.ones(N), P, oracle.first_rows_regular(N, P),
                           oracle.make_settings(max_iters=80, tol=1e-7, **kw))
        print(kw, P, r["converged"], r["iter_count"])
x, it, rn = oracle.gmres(rp, col, val, np.ones(N), None, 3, 1e-9, 200, 7)
f = oracle.cholesky(rp, col, val, False); y = oracle.direct_solve(f, np.ones(N))
print("ok", it)
