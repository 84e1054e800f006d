#!/usr/bin/env python3
"""Ablation builds of the plain-CSR stream kernel (variants 80 + bits: 1 no y store, 2 no x gather, 4 no row
pointers) and its grid size, 256^3, alternating with the full kernel."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the measurement variants live in the measurement build of the library (make -C schwarz-lib_amd probes)
os.environ.setdefault("SCHWZ_HIP_LIB", os.path.join(ROOT, "schwarz-lib_amd", "lib", "libschwz_hip_probes.so"))
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import schwz_amd as S
from spmv_probe import timeit
shp = tuple(int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "256,256,256").split(","))
variants = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "80,81,82,83,84,87,9,6").split(",")]
stream = torch.cuda.current_stream().cuda_stream
prob = S.Problem.laplacian(3, *shp)
sd = S.Subdomain(prob, 1, 0, 2, S.partition_regular(prob.N, 1))
rp, col, val = sd.local_matrix()
A = S.Csr(rp, col, val)
x = torch.randn(prob.N, dtype=torch.float64, device="cuda")
y = torch.zeros(prob.N, dtype=torch.float64, device="cuda")
res = {}
for rep in range(2):
    for v in variants:
        ms = timeit(torch, lambda: A.spmv(x.data_ptr(), y.data_ptr(), 1.0, 0.0, v, stream), 20)
        res.setdefault(v, []).append(round(ms, 4))
print(json.dumps({"shape": shp, "grid": os.environ.get("SCHWZ_STREAM_GRID", "1280"), "ms": res}))
