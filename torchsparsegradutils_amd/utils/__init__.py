"""Solver loops and CSR helpers (names mirror reference ``torchsparsegradutils/utils/__init__.py``)."""

from .bicgstab import BICGSTABSettings, bicgstab
from .linear_cg import LinearCGSettings, linear_cg
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
    "convert_coo_to_csr_indices_values",
    "convert_coo_to_csr",
    "sparse_block_diag",
    "sparse_block_diag_split",
    "stack_csr",
    "sparse_eye",
]
