import os, ctypes as C, numpy as np, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import oracle
L = C.CDLL(os.environ.get("SCHWZ_ASAN_HOST", "/tmp/libhost_asan.so"))
i64, vp = C.c_int64, C.c_void_p
def p(a): return a.ctypes.data_as(vp)
rp, col, val = oracle.laplacian3d(12, 11, 10)
N = len(rp) - 1
# problem from csr + subdomain setup for several P / overlaps
L.schwz_problem_from_csr.argtypes = [i64, vp, vp, vp, C.POINTER(vp)]
rp64 = rp.astype(np.int64)
prob = vp()
rc = L.schwz_problem_from_csr(N, p(rp64), p(col), p(val), C.byref(prob)); assert rc == 0, rc
L.schwz_subdomain_setup.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
pass
for P in (1, 3, 8):
    fr = oracle.first_rows_regular(N, P).astype(np.int64)
    for ov in (2, 4):
        for me in range(P):
            sd = vp()
            rc = L.schwz_subdomain_setup(prob, P, me, ov, p(fr), C.byref(sd)); assert rc == 0, (rc, P, me)
            pass
# factorizations
out7 = [vp() for _ in range(7)]
L.schwz_cholesky.argtypes = [i64, vp, vp, vp, C.c_int] + [C.POINTER(vp)] * 7
assert L.schwz_cholesky(N, p(rp), p(col), p(val), 0, *[C.byref(o) for o in out7]) == 0
out6 = [vp() for _ in range(6)]
L.schwz_ilu0.argtypes = [i64, vp, vp, vp] + [C.POINTER(vp)] * 6
assert L.schwz_ilu0(N, p(rp), p(col), p(val), *[C.byref(o) for o in out6]) == 0
w = vp()
L.schwz_isai.argtypes = [i64, vp, vp, vp, C.c_int, C.POINTER(vp)]
assert L.schwz_isai(N, out6[0], out6[1], out6[2], 1, C.byref(w)) == 0
assert L.schwz_isai(N, out6[3], out6[4], out6[5], 0, C.byref(w)) == 0
# generated problems and the graph partitioner
L.schwz_problem_laplacian.argtypes = [C.c_int, i64, i64, i64, C.POINTER(vp)]
pr2 = vp(); assert L.schwz_problem_laplacian(2, 40, 40, 1, C.byref(pr2)) == 0
part = np.zeros(1600, dtype=np.uint32)
L.schwz_partition_graph.argtypes = [vp, C.c_int, vp]
assert L.schwz_partition_graph(pr2, 8, p(part)) == 0
print("host asan run ok", np.bincount(part))
