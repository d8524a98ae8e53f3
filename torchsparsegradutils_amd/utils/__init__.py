"""Krylov solver loops and index-format helpers of the hot path.  The exported names are the reference's
(``torchsparsegradutils/utils``): user code imports them from here unchanged."""

import sys as _sys

# fused-kernel solver loops (csrc/krylov.hip, minres.hip, bicgstab.hip) and the LSMR driver around the SpMM kernel
from .linear_cg import linear_cg, LinearCGSettings
from .minres import minres, MINRESSettings
from .bicgstab import bicgstab, BICGSTABSettings
from .lsmr import lsmr

# COO/CSR conversion and batching glue
from .utils import convert_coo_to_csr, convert_coo_to_csr_indices_values
from .utils import sparse_block_diag, sparse_block_diag_split, stack_csr, sparse_eye


def last_solve_info(solver: str = "linear_cg"):
    """Iteration count / stopping outcome / final residual of the calling thread's most recent solve with
    ``solver`` in {"linear_cg", "minres", "bicgstab"} (build extension: the reference only prints, see
    utils/linear_cg.py:273-275)."""
    return _sys.modules[__name__ + "." + solver].last_solve_info()


__all__ = [n for n in dir() if not n.startswith("_") and n not in ("utils", "synthetic")]
