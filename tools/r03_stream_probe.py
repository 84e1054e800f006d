#!/usr/bin/env python3
"""Round-3 screening of the plain-CSR stream kernel: copy-probe forms (schwz_stream_probe modes 0-4) and the
ablation builds (variants 80 + bits, spmv_stream.hip) on the shapes a GPU holds in the BASELINE configurations."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the measurement variants live in the measurement build of the library (make -C schwarz-lib_amd probes)
os.environ.setdefault("SCHWZ_HIP_LIB", os.path.join(ROOT, "schwarz-lib_amd", "lib", "libschwz_hip_probes.so"))
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import schwz_amd as S
from spmv_probe import timeit
stream = torch.cuda.current_stream().cuda_stream
lib, check = S.capi.lib, S.capi.check
if "--no-copy" not in sys.argv:
    big = torch.empty(1 << 27, dtype=torch.float64, device="cuda")  # 1 GiB
    dst = torch.empty_like(big)
    big.normal_()
    out = {}
    for rep in range(2):
        for mode in (0, 1, 2, 3, 4):
            ms = timeit(torch, lambda: check(lib.schwz_stream_probe(big.numel(), mode, big.data_ptr(), dst.data_ptr(), stream)), 20)
            out.setdefault("mode%d" % mode, []).append(round((1 if mode == 1 else 2) * big.numel() * 8 / ms / 1e9, 3))
        ms = timeit(torch, lambda: dst.copy_(big), 20)
        out.setdefault("torch_copy", []).append(round(2 * big.numel() * 8 / ms / 1e9, 3))
    print(json.dumps({"copy_probe_TBps": out}), flush=True)
    del big, dst
    torch.cuda.empty_cache()
shapes = [a for a in sys.argv[1:] if "," in a and "=" not in a] or ["256,256,256", "512,512,64"]
variants = [80, 81, 82, 83, 88, 96, 104, 112, 120, 136, 6]
for a in sys.argv[1:]:
    if a.startswith("v="):
        variants = [int(t) for t in a[2:].split(",")]
for shape in shapes:
    shp = tuple(int(t) for t in shape.split(","))
    prob = S.Problem.laplacian(3, *shp)
    sd = S.Subdomain(prob, 1, 0, 2, S.partition_regular(prob.N, 1))
    rp, col, val = sd.local_matrix()
    A = S.Csr(rp, col, val)
    x = torch.randn(prob.N, dtype=torch.float64, device="cuda")
    y = torch.zeros(prob.N, dtype=torch.float64, device="cuda")
    res = {}
    seqs = [int(t) for a in sys.argv[1:] if a.startswith("seq=") for t in a[4:].split(",")] or [0]
    for rep in range(2):
        for sq in seqs:
            os.environ["SCHWZ_STREAM_SEQ"] = str(sq)   # read by the ablation launcher at every launch
            for v in variants:
                if sq and v < 80:
                    continue
                ms = timeit(torch, lambda: A.spmv(x.data_ptr(), y.data_ptr(), 1.0, 0.0, v, stream), 20)
                res.setdefault(("%d" % v) + ("/seq%d" % sq if sq else ""), []).append(round(ms, 4))
    os.environ["SCHWZ_STREAM_SEQ"] = "0"
    print(json.dumps({"shape": shp, "alg_bytes": A.algorithmic_bytes(), "ms": res,
                      "frac_v6": A.algorithmic_bytes() / min(res.get(6, [1e9])) / 1e6 / 8000.0}), flush=True)
    del A, x, y, sd, prob, rp, col, val
    torch.cuda.empty_cache()
