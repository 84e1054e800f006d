#!/usr/bin/env python3
"""A/B timing of the SpMV variants and the device-copy ceiling on one GPU.

    python tools/spmv_probe.py [--size 256] [--reps 20]
Prints one JSON line per variant: average launch ms (torch.cuda events on the
current stream, where the kernels are launched) and algorithmic GB/s.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the measurement variants live in the measurement build of the library (make -C schwarz-lib_amd probes)
os.environ.setdefault("SCHWZ_HIP_LIB", os.path.join(ROOT, "schwarz-lib_amd", "lib", "libschwz_hip_probes.so"))
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))


def timeit(torch, fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--shape", default=None, help="nx,ny,nz instead of a cube")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--variants", default="0,7,6")
    ap.add_argument("--only-spmv", action="store_true")
    a = ap.parse_args()
    import torch
    import schwz_amd as S
    n = a.size
    shp = [int(t) for t in a.shape.split(",")] if a.shape else [n, n, n]
    prob = S.Problem.laplacian(3, *shp)
    sd = S.Subdomain(prob, 1, 0, 2, S.partition_regular(prob.N, 1))
    rp, col, val = sd.local_matrix()
    A = S.Csr(rp, col, val)
    N = prob.N
    x = torch.randn(N, dtype=torch.float64, device="cuda")
    y = torch.zeros(N, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    if a.only_spmv:
        for v in [int(t) for t in a.variants.split(",")]:
            ms = timeit(torch, lambda: A.spmv(x.data_ptr(), y.data_ptr(), 1.0, 0.0, v, stream), a.reps)
            print(json.dumps({"kernel": "spmv variant %d" % v, "ms": ms,
                              "alg_GB/s": A.algorithmic_bytes() / ms / 1e6}))
        return
    # device copy ceiling (read + write)
    big = torch.empty(1 << 28, dtype=torch.float64, device="cuda")  # 2 GiB
    dst = torch.empty_like(big)
    ms = timeit(torch, lambda: dst.copy_(big), a.reps)
    print(json.dumps({"kernel": "torch copy 2GiB", "ms": ms, "GB/s": 2 * big.numel() * 8 / ms / 1e6}))
    for mode, name, factor in ((0, "schwz stream copy 2GiB (double2)", 2), (1, "schwz stream read 2GiB (double2)", 1)):
        ms = timeit(torch, lambda: S.capi.check(S.capi.lib.schwz_stream_probe(
            big.numel(), mode, big.data_ptr(), dst.data_ptr(), stream)), a.reps)
        print(json.dumps({"kernel": name, "ms": ms, "GB/s": factor * big.numel() * 8 / ms / 1e6}))
    del big, dst
    for v in [int(t) for t in a.variants.split(",")]:
        ms = timeit(torch, lambda: A.spmv(x.data_ptr(), y.data_ptr(), 1.0, 0.0, v, stream), a.reps)
        print(json.dumps({"kernel": "spmv variant %d" % v, "rows": N, "nnz": A.nnz, "ms": ms,
                          "alg_GB/s": A.algorithmic_bytes() / ms / 1e6,
                          "frac_of_8TBs": A.algorithmic_bytes() / ms / 1e6 / 8000}))
    # one CG iteration cost by kernel class
    for v in [int(t) for t in a.variants.split(",") if t not in ("1", "2", "3", "5")]:
        sd2 = S.Subdomain(prob, 1, 0, 2, S.partition_regular(prob.N, 1))
        import numpy as np
        sd2.to_device(np.ones(N), precond=S.capi.PRECOND_JACOBI, local_tol=0.0, local_max_iters=20,
                      spmv_variant=v)
        ms = timeit(torch, lambda: sd2.local_solve(stream), 5)
        print(json.dumps({"kernel": "pcg 20 iterations, spmv variant %d" % v, "ms": ms,
                          "ms_per_cg_iteration": ms / 21.0,
                          "alg_GB/s_per_iteration": sd2.algorithmic_bytes(1) / (ms / 21.0) / 1e6}))
        sd2.close()


if __name__ == "__main__":
    main()
