"""``sparse_generic_lstsq`` / ``SparseGenericLstsq`` — drop-in for reference ``torchsparsegradutils/sparse_lstsq.py``
(SURVEY §8 f-2): sparse linear least squares :math:`\\min_x \\|A x - B\\|_2` with sparsity-preserving gradients
(Golub & Pereyra 1973, eq. 4.12, for tall full-column-rank ``A``).

Same signature, same autograd contract (gradient of ``A`` at its stored entries only, in A's layout with A's index
tensors; backward raises ``ValueError`` for a wide ``A``, reference :205-206).  The arithmetic runs on the gfx950
kernels:

=======================================================  ==========================================================
reference (ATen)                                         here
=======================================================  ==========================================================
per-column Python loop over ``lsmr`` (:124-147)          lock-step multi-RHS ``utils.lsmr`` around K1 (``A·``, ``Aᵀ·``)
row expansion + 4 nnz×p gathers + 2 products + 2 sums    ONE K3 SDDMM over 2p columns:
(:229-262)                                               ``<[-G_B | r]_i , [x | A⁺G_B]_j>`` at the stored (i, j)
``A @ x`` for the residual (:246)                        K1 SpMM
=======================================================  ==========================================================
"""

from __future__ import annotations

from typing import Callable, Optional, cast

import torch

from . import _backend as _be
from . import _ops
from . import _pattern as _pt


def sparse_generic_lstsq(
    A: torch.Tensor,
    B: torch.Tensor,
    lstsq: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
    transpose_lstsq: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
) -> torch.Tensor:
    r"""Least-squares solution of :math:`A x \approx B` for sparse tall ``A`` (COO/CSR ``(m, n)``, ``m > n``, full
    column rank) and dense ``B`` (``(m,)`` or ``(m, k)``), differentiable in both.  ``lstsq(A, B) -> X`` and
    ``transpose_lstsq(A, G) -> (Aᵀ)⁺ G`` default to LSMR (mirrors reference ``sparse_lstsq.py:6-153``)."""
    if lstsq is None or transpose_lstsq is None:
        from .utils.lsmr import lsmr

        if lstsq is None:

            def lstsq(AA, BB):  # all right-hand sides in one lock-step LSMR run
                return lsmr(AA, BB)[0]

        if transpose_lstsq is None:

            def transpose_lstsq(AA, BB):  # min ‖Aᵀ y − g‖: the operator is Aᵀ, its adjoint is A itself
                from .utils.lsmr import _transposed_operator

                op, rmat = _transposed_operator(AA)
                return lsmr(rmat, BB, Armat=op, n=AA.shape[0])[0]

    return cast(torch.Tensor, SparseGenericLstsq.apply(A, B, lstsq, transpose_lstsq))


class SparseGenericLstsq(torch.autograd.Function):
    """Autograd kernel behind :func:`sparse_generic_lstsq` (mirrors reference ``sparse_lstsq.py:156-271``)."""

    @staticmethod
    def forward(ctx, A, B, lstsq, transpose_lstsq):
        grad_flag = A.requires_grad or B.requires_grad
        ctx.lstsq = lstsq
        ctx.transpose_lstsq = transpose_lstsq

        x = lstsq(A.detach(), B.detach())
        x.requires_grad = grad_flag
        if B.dim() == 1:
            if x.dim() == 2:
                x = x.squeeze()
        elif x.dim() == 1:
            x = x.unsqueeze(1)

        ctx.save_for_backward(A.detach(), B.detach(), x.detach())
        return x

    @staticmethod
    def backward(ctx, grad):  # type: ignore[override]
        A, B, x = ctx.saved_tensors
        if B.ndim == 1:
            B = B.unsqueeze(1)
        if x.ndim == 1:
            x = x.unsqueeze(1)

        # gradB = (Aᵀ)⁺ grad   (reference :198-200)
        gradB = ctx.transpose_lstsq(A, grad)
        if gradB.ndim == 1:
            gradB = gradB.unsqueeze(1)
        if A.shape[1] > A.shape[0]:  # reference :205-206
            raise ValueError(f"A should be a tall full-rank matrix. Got A.shape={A.shape}")

        # gradA[i,j] = -<gradB[i,:], x[j,:]> + <(B - A x)[i,:], (A⁺ gradB)[j,:]>   (reference :229-262) — both terms in
        # one SDDMM over the concatenated columns, no row expansion and no nnz×p temporaries
        if A.layout == torch.sparse_coo:
            Ac = A if A.is_coalesced() else A.coalesce()
            idx = Ac.indices()
            plan, values = _pt.from_coo_2d(idx, A.shape, coalesced=True), Ac.values()
        else:
            idx = None
            plan, values = _pt.from_csr(A), A.values()
        _be.require_device(values, B, x)
        residual = B - _ops.spmm(plan, values, x.contiguous())
        Apgb = ctx.lstsq(A, gradB)
        if Apgb.dim() == 1:
            Apgb = Apgb.unsqueeze(1)
        left = torch.cat((-gradB, residual), dim=1).contiguous()
        right = torch.cat((x, Apgb), dim=1).contiguous()
        gvals = _ops.sddmm(plan, left, right)
        if idx is not None:
            gradA = torch.sparse_coo_tensor(idx, gvals, A.shape)
        else:
            gradA = torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), gvals, A.shape)

        if grad.ndim == 1:
            gradB = gradB.squeeze()
        return gradA, gradB, None, None
