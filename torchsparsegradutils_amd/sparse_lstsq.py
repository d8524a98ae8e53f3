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


def _columns(t: torch.Tensor) -> torch.Tensor:
    """(m,) -> (m, 1); matrices pass through."""
    return t.unsqueeze(1) if t.ndim == 1 else t


class SparseGenericLstsq(torch.autograd.Function):
    """Autograd kernel behind :func:`sparse_generic_lstsq` (same contract as reference ``sparse_lstsq.py:156-271``:
    solution shaped like ``B``, gradient of ``A`` on A's pattern in A's layout, ``ValueError`` for a wide ``A`` in
    backward)."""

    @staticmethod
    def forward(ctx, A, B, lstsq, transpose_lstsq):
        ctx.solvers = (lstsq, transpose_lstsq)
        wants_grad = A.requires_grad or B.requires_grad
        A_, B_ = A.detach(), B.detach()
        sol = lstsq(A_, B_)
        # the solution has the rank of B, whatever the user's solver returned (reference :178-185)
        if B_.ndim == 1 and sol.ndim == 2:
            sol = sol.squeeze()
        elif B_.ndim != 1 and sol.ndim == 1:
            sol = sol.unsqueeze(1)
        sol.requires_grad = wants_grad
        ctx.save_for_backward(A_, B_, sol.detach())
        return sol

    @staticmethod
    def backward(ctx, grad):  # type: ignore[override]
        A, B, x = ctx.saved_tensors
        lstsq, transpose_lstsq = ctx.solvers
        vector_rhs = grad.ndim == 1
        B, x = _columns(B), _columns(x)

        grad_b = _columns(transpose_lstsq(A, grad))              # (Aᵀ)⁺ grad   (reference :198-200)
        m_rows, n_cols = A.shape
        if n_cols > m_rows:                                      # reference :205-206
            raise ValueError(f"A should be a tall full-rank matrix. Got A.shape={A.shape}")

        # Golub–Pereyra 4.12 restricted to A's pattern (reference :229-262):
        #   gradA[i,j] = −<grad_b[i,:], x[j,:]> + <(B − A x)[i,:], (A⁺ grad_b)[j,:]>
        # both inner products in ONE SDDMM over the concatenated columns: no row expansion, no nnz×p temporaries
        coo_index = None
        if A.layout == torch.sparse_coo:
            Ac = A if A.is_coalesced() else A.coalesce()
            coo_index = Ac.indices()
            plan, values = _pt.from_coo_2d(coo_index, A.shape, coalesced=True), Ac.values()
        else:
            plan, values = _pt.from_csr(A), A.values()
        _be.operand_device(values, B, x)
        residual = B - _ops.spmm(plan, values, x.contiguous())
        pinv_gb = _columns(lstsq(A, grad_b))
        row_side = torch.cat((-grad_b, residual), dim=1).contiguous()
        col_side = torch.cat((x, pinv_gb), dim=1).contiguous()
        grad_vals = _ops.sddmm(plan, row_side, col_side)
        if coo_index is not None:
            grad_a = torch.sparse_coo_tensor(coo_index, grad_vals, A.shape)
        else:
            grad_a = torch.sparse_csr_tensor(A.crow_indices(), A.col_indices(), grad_vals, A.shape)
        return grad_a, (grad_b.squeeze() if vector_rhs else grad_b), None, None
