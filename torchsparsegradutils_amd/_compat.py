"""``linalg_solve_triangular_compat`` — drop-in for reference ``torchsparsegradutils/_compat.py:8-48`` (the reference's own
benchmarks import it directly: ``benchmarks/sparse_triangular_solve_rand.py:30,71-72``).

Dense coefficient matrices go to ``torch.linalg.solve_triangular`` exactly as the reference sends them (``transpose=True`` is
folded into a transposed view with ``upper`` flipped, ``_compat.py:25-27``).  Sparse COO / CSR matrices — where the reference
keeps the single legacy ``torch.triangular_solve`` call (``_compat.py:42-48``) — are solved by the K4 sync-free sweep
(``tsgu_csr_sptrsm``): same flag meaning (entries of the other triangle are ignored, ``unitriangular`` ignores stored
diagonals, ``transpose`` solves with ``Aᵀ``), no autograd graph, like the legacy op's ``.solution`` on a sparse operand.
"""

from __future__ import annotations

import torch


def linalg_solve_triangular_compat(
    A: torch.Tensor,
    B: torch.Tensor,
    *,
    upper: bool,
    unitriangular: bool = False,
    transpose: bool = False,
) -> torch.Tensor:
    """Solve a triangular system with the dense (torch) or sparse (HIP) backend; signature of reference ``_compat.py:8-15``."""
    if A.layout == torch.strided:
        if transpose:
            A = A.transpose(-2, -1)
            upper = not upper
        return torch.linalg.solve_triangular(A, B, upper=upper, unitriangular=unitriangular)

    from .sparse_solve import _solve, _TriOperand

    if A.layout not in (torch.sparse_coo, torch.sparse_csr):
        raise ValueError("A should be in either COO or CSR sparse format")
    # the legacy op checks its operands' shapes itself (reference _compat.py:42-48 hands them to torch.triangular_solve); the sweep
    # takes raw pointers, so the same conditions are checked here, in sparse_triangular_solve's words (sparse_solve.py:131-146)
    if A.dim() not in (2, 3) or A.dim() != B.dim():
        raise ValueError("A and B must both be 2D or both be 3D tensors")
    if B.layout != torch.strided:
        raise ValueError("B must be a dense (strided) tensor")
    if A.shape[-2] != A.shape[-1]:
        raise ValueError("A must be square on its last two dimensions")
    if A.size(-1) != B.size(-2):
        raise ValueError(f"Incompatible inner dimensions: A[..., {A.size(-2)}] vs B[..., {B.size(-2)}]")
    if A.dim() == 3 and A.size(0) != B.size(0):
        raise ValueError("If batched, A and B must have the same batch size")
    A, B = A.detach(), B.detach()
    if A.device != B.device:
        raise RuntimeError(f"A and B must be on the same device, got {A.device} and {B.device}")
    op = _TriOperand(A)
    batched = B.dim() == 3
    rhs = B.reshape(-1, B.size(-1)) if batched else B
    x = _solve(op.plan, op.values, rhs, upper, unitriangular, transpose)
    return x.view(B.shape) if batched else x
