"""schwz_amd -- MI355X-native Restricted Additive Schwarz hot path.

Host-side mirror of the schwarz-lib solver API (Settings / Metadata / SolverRAS)
over libschwz_hip.so (include/schwz_hip.h).  Importing the package loads the
shared library and fails loudly if it has not been built.
"""
from . import _capi as capi  # noqa: F401  (loads libschwz_hip.so)
from ._capi import SchwzError, NotImplementedSchwz  # noqa: F401
from .comm import InProcessComm, TorchDistComm, WindowComm  # noqa: F401
from .core import (Csr, DeviceWindow, Gmres, Pcg, PeerWindow, Problem, Subdomain, Trs, cholesky, gather, ilu0, isai, scatter,  # noqa: F401
                   partition_regular, partition_regular2d, rhs_random)
from .solver import (HipBackend, Metadata, Settings, SolverRAS,  # noqa: F401
                     PARTITION_CUSTOM, PARTITION_METIS, PARTITION_REGULAR, PARTITION_REGULAR2D,
                     SOLVER_DIRECT_CHOLMOD, SOLVER_DIRECT_GINKGO, SOLVER_DIRECT_UMFPACK,
                     SOLVER_ITERATIVE_GINKGO)

__all__ = ["Settings", "Metadata", "SolverRAS", "HipBackend", "InProcessComm", "TorchDistComm", "WindowComm",
           "Problem", "Subdomain", "Csr", "Pcg", "Trs", "SchwzError"]
