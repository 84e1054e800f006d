#!/usr/bin/env python3
"""Plain-CSR SpMV: the straight-line pipeline (variant 6, spmv_stream.hip) against spmv_tiled2_kernel (variant 9):
bit identity of y on a few shapes, then timing at 256^3 and on the 512 x 512 x 64 slab."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))
import torch
import schwz_amd as S
sys.path.insert(0, os.path.join(ROOT, "tools"))
from spmv_probe import timeit

stream = torch.cuda.current_stream().cuda_stream
for shp in ((24, 20, 16), (64, 64, 64), (100, 37, 29), (256, 256, 256), (512, 512, 64)):
    prob = S.Problem.laplacian(3, *shp)
    sd = S.Subdomain(prob, 1, 0, 2, S.partition_regular(prob.N, 1))
    rp, col, val = sd.local_matrix()
    A = S.Csr(rp, col, val)
    N = prob.N
    x = torch.randn(N, dtype=torch.float64, device="cuda")
    y6 = torch.zeros(N, dtype=torch.float64, device="cuda")
    y9 = torch.zeros(N, dtype=torch.float64, device="cuda")
    A.spmv(x.data_ptr(), y6.data_ptr(), 1.0, 0.0, 6, stream)
    A.spmv(x.data_ptr(), y9.data_ptr(), 1.0, 0.0, 9, stream)
    torch.cuda.synchronize()
    same = bool(torch.equal(y6, y9))
    out = {"shape": shp, "bit_identical": same, "max_abs_diff": float((y6 - y9).abs().max())}
    if N >= 1 << 24:
        for v in (9, 6, 9, 6):
            ms = timeit(torch, lambda: A.spmv(x.data_ptr(), y6.data_ptr(), 1.0, 0.0, v, stream), 20)
            out.setdefault("ms_v%d" % v, []).append(round(ms, 4))
        out["frac_v6"] = A.algorithmic_bytes() / min(out["ms_v6"]) / 1e6 / 8000.0
        out["frac_v9"] = A.algorithmic_bytes() / min(out["ms_v9"]) / 1e6 / 8000.0
    print(json.dumps(out), flush=True)
    del A, x, y6, y9, sd, prob
    torch.cuda.empty_cache()
