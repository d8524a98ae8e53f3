"""CPU tensors: the package's own torch-op path.

BASELINE ``configs[0]`` is "sparse_mm COO 4096×4096 … on CPU (plumbing, no GPU)" and the reference runs on CPU by default
(``sparse_matmul.py:141-163``; its tests use ``DEVICES=[torch.device("cpu")]``).  Operands that live on the CPU are therefore
computed here, with the ATen calls the reference itself makes at the cited lines — and ONLY such operands: the switch is the
device of the tensors, nothing else.  A tensor on the GPU never reaches this module (every function refuses one), and a missing
``libtsgu_hip.so`` is an error for GPU operands, not a reason to come here: this is the reference's own CPU behaviour for CPU
callers, not a fallback of the MI355X path.  Nothing here touches ``oracle/`` (test infrastructure).

=========================================  ====================================================================
reference (file:line)                      here
=========================================  ====================================================================
``torch.sparse.mm(A, B)``  :155            :func:`spmm` on the cached 2-D (block-diagonal if batched) pattern
gathers · mul · sum  :186-205              :func:`sddmm`, in entry chunks: the nnz×p temporaries never exist
``torch.sparse.mm(A.t(), G)``  :229        :func:`spmm` on the cached transposed pattern
``torch.triangular_solve``  _compat:42-48  :func:`sptrsm` (the same call, same flags)
column dots of the Krylov loops            :func:`coldot`
=========================================  ====================================================================
"""

from __future__ import annotations

import torch

from . import _pattern as _pt

# entries per chunk of the masked product: chunk × p elements of temporaries (two gathers and a product) at a time
_SDDMM_CHUNK_ELEMS = 1 << 22


def _cpu_only(*tensors) -> None:
    for t in tensors:
        if t is not None and t.is_cuda:
            raise RuntimeError("torchsparsegradutils_amd._cpu was handed a GPU tensor: the torch-op path serves CPU operands only")


def _flat(plan: _pt.RowGather) -> _pt.RowGather:
    return _pt.flat_of(plan) if plan.batch is not None else plan


def _values_in_plan_order(plan: _pt.RowGather, values: torch.Tensor) -> torch.Tensor:
    v = values.reshape(-1)
    return v if plan.perm is None else v.index_select(0, plan.perm.reshape(-1).to(torch.int64))


def matrix(plan: _pt.RowGather, values: torch.Tensor) -> torch.Tensor:
    """The 2-D torch CSR tensor of (plan, values): A's own index arrays (int32 stays int32), block diagonal for a batched plan —
    what the reference assembles with ``sparse_block_diag`` for every batched input (sparse_matmul.py:151-153)."""
    f = _flat(plan)
    return torch.sparse_csr_tensor(f.crow, f.col, _values_in_plan_order(f, values), (f.n_rows, f.n_cols))


def spmm(plan: _pt.RowGather, values: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A·B (reference sparse_matmul.py:155); with a transposed plan (``perm`` into A's values) it is Aᵀ·G (:229)."""
    _cpu_only(values, B, plan.crow)
    p = B.size(-1)
    out = torch.sparse.mm(matrix(plan, values), B.reshape(-1, p))
    return out.view(B.shape[:-2] + (plan.n_rows, p))


def sddmm(plan: _pt.RowGather, G: torch.Tensor, B: torch.Tensor, alpha: float = 1.0, swap_roles: bool = False) -> torch.Tensor:
    """alpha·<G[row k], B[col k]> (roles swapped: <B[row k], G[col k]>) at the plan's entries, in plan order, shaped like the
    plan's column array (reference sparse_matmul.py:186-205, sparse_solve.py:216-235)."""
    _cpu_only(G, B, plan.crow)
    f = _flat(plan)
    p = G.size(-1)
    row_side = (B if swap_roles else G).reshape(-1, p)
    col_side = (G if swap_roles else B).reshape(-1, p)
    rows, cols = f.row_indices().reshape(-1).to(torch.int64), f.col.reshape(-1).to(torch.int64)
    nnz = cols.numel()
    out = torch.empty(nnz, dtype=torch.result_type(G, B), device=G.device)
    step = max(1, _SDDMM_CHUNK_ELEMS // max(p, 1))
    for s in range(0, nnz, step):
        e = min(nnz, s + step)
        torch.sum(row_side.index_select(0, rows[s:e]) * col_side.index_select(0, cols[s:e]), dim=-1, out=out[s:e])
    if alpha != 1.0:
        out.mul_(alpha)
    return out.view(plan.col.shape)


def coo_sddmm(rows: torch.Tensor, cols: torch.Tensor, G: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """The same product at explicit (row, column) pairs: un-coalesced COO, one gradient entry per stored duplicate
    (reference sparse_matmul.py:185,201-205)."""
    _cpu_only(rows, cols, G, B)
    nnz, p = rows.numel(), G.size(-1)
    out = torch.empty(nnz, dtype=torch.result_type(G, B), device=G.device)
    step = max(1, _SDDMM_CHUNK_ELEMS // max(p, 1))
    for s in range(0, nnz, step):
        e = min(nnz, s + step)
        torch.sum(G.index_select(0, rows[s:e]) * B.index_select(0, cols[s:e]), dim=-1, out=out[s:e])
    return out


def sptrsm(plan: _pt.RowGather, values: torch.Tensor, rhs: torch.Tensor, upper: bool, unit: bool, transpose: bool) -> torch.Tensor:
    """X = op(A)⁻¹·rhs by the legacy ATen call the reference keeps for sparse operands (_compat.py:42-48), same flags."""
    _cpu_only(values, rhs, plan.crow)
    if plan.n_rows == 0 or rhs.size(-1) == 0:
        return torch.empty_like(rhs)
    return torch.triangular_solve(rhs.contiguous(), matrix(plan, values), upper=upper, transpose=transpose, unitriangular=unit).solution


def coldot(X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
    """Column-wise dot products of two (n, p) arrays -> (p,)."""
    _cpu_only(X, Y)
    return (X * Y).sum(dim=0)
