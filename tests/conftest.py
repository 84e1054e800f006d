import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "schwarz-lib_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def schwz():
    import schwz_amd
    return schwz_amd


@pytest.fixture(scope="session")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    return torch


def convection_diffusion_2d(n, cx=6.0, cy=-3.0):
    """Non-symmetric test matrix: 5-point diffusion plus first-order upwind convection on an
    n x n grid (natural ordering).  Returns int32 CSR arrays with sorted columns."""
    import numpy as np
    import scipy.sparse as sp
    eye = sp.identity(n, format="csr")
    t = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n), format="csr")
    d = sp.diags([-1.0, 1.0], [-1, 0], shape=(n, n), format="csr")
    a = (sp.kron(eye, t) + sp.kron(t, eye) + 0.5 * cx * sp.kron(eye, d) - 0.5 * cy * sp.kron(d.T, eye)).tocsr()
    a.eliminate_zeros()
    a.sort_indices()
    return a.indptr.astype(np.int32), a.indices.astype(np.int32), a.data.astype(np.float64)


@pytest.fixture(scope="session")
def convdiff():
    return convection_diffusion_2d
