"""Generates the golden fixtures under tests/golden/ (run once, here, by hand).

The reference ships no tests and no expected outputs (TESTING.md:1-2), so the
fixtures are minted from an INDEPENDENT solver: scipy's sparse direct solve
(SuperLU) on the same inputs.  Inputs taken from the reference are data files
only (matrices/ani3_crop.mtx, matrices/ani4_crop.mtx), stored here as CSR
arrays.  Nothing in this script runs or imports reference code.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import scipy.io
import scipy.sparse as sp
import scipy.sparse.linalg as spl

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/matrices"


def lap2d(n):
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    I = sp.identity(n)
    return (sp.kron(I, T) + sp.kron(T, I)).tocsr()


def lap3d(nx, ny, nz):
    def T(n):
        return sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    Ix, Iy, Iz = sp.identity(nx), sp.identity(ny), sp.identity(nz)
    return (sp.kron(Iz, sp.kron(Iy, T(nx))) + sp.kron(Iz, sp.kron(T(ny), Ix)) +
            sp.kron(T(nz), sp.kron(Iy, Ix))).tocsr()


def solve_ones(A):
    b = np.ones(A.shape[0])
    x = spl.spsolve(A.tocsc(), b)
    r = b - A @ x
    return x, float(np.linalg.norm(r) / np.linalg.norm(b))


def main():
    meta = {}
    for name in ("ani3_crop", "ani4_crop"):
        A = scipy.io.mmread(os.path.join(REF, name + ".mtx")).tocsr()
        A.sort_indices()
        x, rel = solve_ones(A)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rp=A.indptr.astype(np.int32),
                            col=A.indices.astype(np.int32), val=A.data.astype(np.float64),
                            x_ones=x)
        meta[name] = dict(n=int(A.shape[0]), nnz=int(A.nnz), x_norm=float(np.linalg.norm(x)),
                          direct_rel_residual=rel,
                          symmetric_defect=float(abs(A - A.T).max()))
    for n in (16, 64):
        A = lap2d(n)
        x, rel = solve_ones(A)
        np.savez_compressed(os.path.join(HERE, "lap2d_%d.npz" % n), x_ones=x)
        meta["lap2d_%d" % n] = dict(n=n * n, nnz=int(A.nnz), x_norm=float(np.linalg.norm(x)),
                                    x_max=float(x.max()), direct_rel_residual=rel)
    for shape in ((12, 12, 12), (16, 10, 7)):
        A = lap3d(*shape)
        x, rel = solve_ones(A)
        tag = "lap3d_%dx%dx%d" % shape
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), x_ones=x)
        meta[tag] = dict(n=int(A.shape[0]), nnz=int(A.nnz), x_norm=float(np.linalg.norm(x)),
                         x_max=float(x.max()), direct_rel_residual=rel)
    # scipy CSR of the generators themselves (pins the oracle's generators)
    A = lap2d(5)
    A.eliminate_zeros()
    A.sort_indices()
    meta["lap2d_5_csr"] = dict(rp=A.indptr.tolist(), col=A.indices.tolist(), val=A.data.tolist())
    A = lap3d(3, 2, 2)
    A.eliminate_zeros()
    A.sort_indices()
    meta["lap3d_3x2x2_csr"] = dict(rp=A.indptr.tolist(), col=A.indices.tolist(),
                                   val=A.data.tolist())
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    sys.exit(main())
