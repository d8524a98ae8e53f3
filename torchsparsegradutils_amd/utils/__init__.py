"""Solver loops and CSR helpers (names mirror reference ``torchsparsegradutils/utils/__init__.py``)."""

from .bicgstab import BICGSTABSettings, bicgstab
from .linear_cg import LinearCGSettings, linear_cg
from .lsmr import lsmr
from .minres import MINRESSettings, minres
from .utils import (
    convert_coo_to_csr,
    convert_coo_to_csr_indices_values,
    sparse_block_diag,
    sparse_block_diag_split,
    sparse_eye,
    stack_csr,
)

__all__ = [
    "linear_cg",
    "LinearCGSettings",
    "minres",
    "MINRESSettings",
    "bicgstab",
    "BICGSTABSettings",
    "lsmr",
    "convert_coo_to_csr_indices_values",
    "convert_coo_to_csr",
    "sparse_block_diag",
    "sparse_block_diag_split",
    "stack_csr",
    "sparse_eye",
]


def last_solve_info(solver: str = "linear_cg"):
    """Iteration count / stopping outcome / final residual of the calling thread's most recent solve with
    ``solver`` in {"linear_cg", "minres", "bicgstab"} (build extension: the reference only prints, see
    utils/linear_cg.py:273-275)."""
    import sys as _sys

    mod = _sys.modules[__name__ + "." + solver]
    return mod.last_solve_info()
