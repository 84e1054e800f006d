#!/usr/bin/env python3
"""A few launches of the plain-CSR SpMV (variant 6, spmv_stream_kernel) on one z-slab of a 3-D Poisson grid: the
program tools/profile_plain.sh puts under rocprofv3 (kernel trace / FETCH_SIZE / WRITE_SIZE passes per shape).

    python tools/plain_spmv_run.py 512,512,64 [launches]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))
for k in ("SCHWZ_SPMV_PAIR", "SCHWZ_SPMV_PATTERN", "SCHWZ_SPMV_DICT"):
    os.environ[k] = "0"   # plain CSR only: no coded copies of the matrix are built
import torch
import schwz_amd as S
shape = tuple(int(t) for t in sys.argv[1].split(","))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
prob = S.Problem.laplacian(3, *shape)
sd = S.Subdomain(prob, 1, 0, 2, S.partition_regular(prob.N, 1))
rp, col, val = sd.local_matrix()
A = S.Csr(rp, col, val)
n = len(rp) - 1
del rp, col, val
x = torch.randn(n, dtype=torch.float64, device="cuda")
y = torch.zeros(n, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    A.spmv(x.data_ptr(), y.data_ptr(), 1.0, 0.0, 6, stream)
torch.cuda.synchronize()
print("done", shape, n, A.algorithmic_bytes())
