"""MI355X-native (gfx950) implementation of torchsparsegradutils' sparse hot path.

Drop-in names for ``sparse_mm`` / ``sparse_triangular_solve`` / ``sparse_generic_solve``
(reference ``torchsparsegradutils/__init__.py:1-16``); the arithmetic runs in hand-written HIP
kernels behind the C ABI in ``include/tsgu_hip.h``.  GPU only — there is no CPU fallback.
"""

from ._backend import poll_errors
from ._compat import linalg_solve_triangular_compat
from ._pattern import wait_for_plans
from .sparse_lstsq import SparseGenericLstsq, sparse_generic_lstsq
from .sparse_matmul import SparseMatMul, sparse_mm
from .sparse_solve import (
    SparseGenericSolve,
    SparseTriangularSolve,
    sparse_generic_solve,
    sparse_triangular_solve,
)

__all__ = [
    "sparse_mm",
    "sparse_triangular_solve",
    "sparse_generic_solve",
    "sparse_generic_lstsq",
    "SparseGenericLstsq",
    "wait_for_plans",
    "poll_errors",
    "SparseMatMul",
    "SparseTriangularSolve",
    "SparseGenericSolve",
    "linalg_solve_triangular_compat",
]

__version__ = "0.1.0"


def _prefetch_lazy_torch_modules() -> None:
    """torch.autograd's Python front end imports torch.fx.experimental.symbolic_shapes (and with it sympy: 150-450 ms)
    the first time a backward pass is given explicit output gradients — inside the first training step of every process.
    Import it now, together with this package, where a start-up cost belongs (TSGU_PREFETCH_IMPORTS=0: off).  On the importing
    thread on purpose: a helper thread would save the time but can meet the main thread's own imports in a lock cycle, which
    Python resolves by handing one of them a half-initialised module."""
    import os

    if os.environ.get("TSGU_PREFETCH_IMPORTS", "1") != "1":
        return
    try:
        import importlib

        importlib.import_module("torch.fx.experimental.symbolic_shapes")
    except Exception:  # noqa: BLE001  (purely an optimisation)
        pass


_prefetch_lazy_torch_modules()
