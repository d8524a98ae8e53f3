"""ATen-op port of the reference's hot path — TEST / BASELINE INFRASTRUCTURE ONLY.

The reference is pure Python over PyTorch ATen ops; on a CPU its cost profile (the two nnz×p
gathers, the CSC→CSR conversion inside ``sparse.mm(A.t(), ·)``, the ≈15-op CG chain) is a
property of exactly that op sequence.  This module restates the sequence, in our own words, so
that ``bench.py``'s ``cpu_baseline`` leg can time "what the reference does on host cores" on the
GPU box, where the reference checkout is not available.  It is also a second checker for tests.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Cited lines are in the reference checkout (torchsparsegradutils/...).
"""

from __future__ import annotations

import warnings

import torch


def mm_forward(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """sparse_matmul.py:155 — one sparse addmm."""
    return torch.sparse.mm(A, B)


def mm_backward(A: torch.Tensor, B: torch.Tensor, G: torch.Tensor):
    """sparse_matmul.py:186-211 and :229 — row-pointer expansion, two nnz×p gathers, product,
    row-sum; then a sparse addmm with the transposed (CSC-viewed) operand."""
    crow, col = A.crow_indices(), A.col_indices()
    n = A.size(0)
    row = torch.repeat_interleave(torch.arange(n, device=A.device), crow[1:] - crow[:-1])
    g_sel = G.index_select(0, row)
    b_sel = B.index_select(0, col)
    grad_vals = (g_sel * b_sel).sum(dim=1)
    gradA = torch.sparse_csr_tensor(crow, col, grad_vals, A.shape)
    gradB = torch.sparse.mm(A.t(), G)
    return gradA, gradB


def tri_forward(A: torch.Tensor, B: torch.Tensor, upper: bool, unit: bool, transpose: bool) -> torch.Tensor:
    """_compat.py:42-48 — legacy sparse triangular solve."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return torch.triangular_solve(B, A, upper=upper, transpose=transpose, unitriangular=unit).solution


def tri_backward(A, x, G, upper, unit, transpose):
    """sparse_solve.py:202-240."""
    gradB = tri_forward(A, G, upper, unit, not transpose)
    crow, col = A.crow_indices(), A.col_indices()
    row = torch.repeat_interleave(torch.arange(A.size(0), device=A.device), crow[1:] - crow[:-1])
    if transpose:
        lhs, rhs = -gradB.index_select(0, col), x.index_select(0, row)
    else:
        lhs, rhs = -gradB.index_select(0, row), x.index_select(0, col)
    vals = (lhs * rhs).sum(dim=1)
    return torch.sparse_csr_tensor(crow, col, vals, A.shape), gradB


def cg_iterations(A: torch.Tensor, rhs: torch.Tensor, n_iter: int, eps: float = 1e-10):
    """n_iter iterations of the un-preconditioned loop, op for op (utils/linear_cg.py:257-266,
    :294, :322, :64-95, :372-374) including the per-iteration host sync of :376-380."""
    e = torch.tensor(eps, dtype=rhs.dtype)
    rhs_norm = torch.linalg.vector_norm(rhs, ord=2, dim=-2, keepdim=True)
    zero_rhs = rhs_norm.lt(e)
    rhs_norm = rhs_norm.masked_fill_(zero_rhs, 1)
    rhs = rhs.div(rhs_norm)
    x = torch.zeros_like(rhs)
    r = rhs - A.matmul(x)
    p = r.clone()
    rr = p.mul(r).sum(-2, keepdim=True)
    rnorm = torch.linalg.vector_norm(r, ord=2, dim=-2, keepdim=True)
    conv = torch.lt(rnorm, 1e-10)
    tmp = torch.empty_like(r)
    alpha = torch.empty_like(rr)
    beta = torch.empty_like(rr)
    is_zero = torch.empty_like(rr, dtype=torch.bool)
    for _ in range(n_iter):
        Ap = A.matmul(p)
        torch.mul(p, Ap, out=tmp)
        torch.sum(tmp, dim=-2, keepdim=True, out=alpha)
        torch.lt(alpha, e, out=is_zero)
        alpha.masked_fill_(is_zero, 1)
        torch.div(rr, alpha, out=alpha)
        alpha.masked_fill_(is_zero, 0)
        alpha.masked_fill_(conv, 0)
        torch.addcmul(r, -alpha, Ap, out=r)
        z = r.clone()
        torch.addcmul(x, alpha, p, out=x)
        beta.resize_as_(rr).copy_(rr)
        torch.mul(r, z, out=tmp)
        torch.sum(tmp, -2, keepdim=True, out=rr)
        torch.lt(beta, e, out=is_zero)
        beta.masked_fill_(is_zero, 1)
        torch.div(rr, beta, out=beta)
        beta.masked_fill_(is_zero, 0)
        p.mul_(beta).add_(z)
        torch.linalg.vector_norm(r, ord=2, dim=-2, keepdim=True, out=rnorm)
        rnorm.masked_fill_(zero_rhs, 0)
        torch.lt(rnorm, 1e-10, out=conv)
        bool(rnorm.mean() < 0.0)  # the reference's per-iteration host read
    return x.mul(rhs_norm)
