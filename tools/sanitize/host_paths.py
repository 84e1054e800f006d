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
# distributed ingest: the rows a subdomain reads, cut out and used as the only row source
L.schwz_subdomain_sizes.argtypes = [vp, vp]
L.schwz_subdomain_local_to_global.argtypes = [vp, vp]
L.schwz_problem_extract_rows.argtypes = [vp, i64, vp, vp, vp, vp]
L.schwz_problem_from_rows.argtypes = [i64, i64, vp, vp, vp, vp, C.POINTER(vp)]
for P, ov in ((3, 2), (5, 3)):
    fr = oracle.first_rows_regular(N, P).astype(np.int64)
    for me in range(P):
        sd = vp()
        assert L.schwz_subdomain_setup(prob, P, me, ov, p(fr), C.byref(sd)) == 0
        sz = np.zeros(10, dtype=np.int64)
        assert L.schwz_subdomain_sizes(sd, p(sz)) == 0
        l2g = np.zeros(sz[1] + sz[3], dtype=np.int64)
        assert L.schwz_subdomain_local_to_global(sd, p(l2g)) == 0
        rows = np.sort(l2g[:sz[1]])
        rpr = np.zeros(len(rows) + 1, dtype=np.int64)
        assert L.schwz_problem_extract_rows(prob, len(rows), p(rows), p(rpr), None, None) == 0
        cr = np.zeros(rpr[-1], dtype=col.dtype); vr = np.zeros(rpr[-1])
        assert L.schwz_problem_extract_rows(prob, len(rows), p(rows), p(rpr), p(cr), p(vr)) == 0
        part_prob = vp()
        assert L.schwz_problem_from_rows(N, len(rows), p(rows), p(rpr), p(cr), p(vr), C.byref(part_prob)) == 0
        sd2 = vp()
        assert L.schwz_subdomain_setup(part_prob, P, me, ov, p(fr), C.byref(sd2)) == 0
        sd3 = vp()  # another rank's subdomain from this part: refused (a row is missing)
        rc3 = L.schwz_subdomain_setup(part_prob, P, (me + 1) % P, ov, p(fr), C.byref(sd3)); assert P != 3 or rc3 != 0, (P, ov, me)
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
