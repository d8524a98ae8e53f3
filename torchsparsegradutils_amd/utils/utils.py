"""CSR / COO structure helpers used either side of the HIP kernels.

Same names, arguments and error behaviour as the reference helpers
(``torchsparsegradutils/utils/utils.py``; each function cites the lines it mirrors), written
as vectorised index arithmetic: no per-block ``.item()`` host syncs, no Python loop over the
batch where the layout allows it.  Pure integer/index plumbing — device agnostic.
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

_INT = (torch.int32, torch.int64)


def stack_csr(tensors: List[torch.Tensor], dim: int = 0) -> torch.Tensor:
    """Stack equally-shaped 2-D CSR tensors (same nnz each) into a batched CSR tensor.

    Mirrors reference ``utils/utils.py:6-88`` (``torch.stack`` does not support CSR).
    """
    if not isinstance(tensors, (list, tuple)):
        raise TypeError("Expected a list of tensors, but got {}.".format(type(tensors)))
    if len(tensors) == 0:
        raise ValueError("Cannot stack empty list of tensors.")
    first = tensors[0]
    if any(t.shape != first.shape for t in tensors):
        raise ValueError("All tensors must have the same shape.")
    if any(t.layout != torch.sparse_csr for t in tensors):
        raise ValueError("All tensors must be in CSR layout.")
    if any(t.ndim != 2 for t in tensors):
        raise ValueError("All tensors must be 2D.")
    parts = [(t.crow_indices(), t.col_indices(), t.values()) for t in tensors]
    crow, col, val = (torch.stack(group, dim=dim) for group in zip(*parts))
    shape = list(first.shape)
    shape.insert(dim, len(tensors))
    return torch.sparse_csr_tensor(crow, col, val, tuple(shape))


def _sort_coo_indices(indices: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Lexicographic sort of COO index columns; returns (sorted indices, permutation).

    Mirrors reference ``utils/utils.py:91-149`` (there: ``torch.unique(dim=-1)`` + ``argsort`` of the
    inverse map, valid for duplicate-free input).  Implemented as a least-significant-key-first
    sequence of stable sorts, which is the same ordering as ``Tensor.coalesce()``.
    """
    perm = torch.arange(indices.shape[-1], device=indices.device)
    for key in range(indices.shape[0] - 1, -1, -1):
        perm = perm[torch.argsort(indices[key][perm], stable=True)]
    return indices[:, perm].contiguous(), perm


def _compress_row_indices(row_indices: torch.Tensor, num_rows: int) -> torch.Tensor:
    """Sorted row indices -> CSR row pointer.  Mirrors reference ``utils/utils.py:152-233``."""
    if not isinstance(row_indices, torch.Tensor):
        raise TypeError("row_indices must be a torch.Tensor.")
    if row_indices.ndim != 1:
        raise ValueError(f"row_indices must be 1D, got shape {tuple(row_indices.shape)}.")
    if row_indices.dtype not in _INT:
        raise TypeError("row_indices must have integer dtype (torch.int32 or torch.int64).")
    if not isinstance(num_rows, int) or num_rows <= 0:
        raise ValueError("num_rows must be a positive integer.")
    if row_indices.numel() > 0:
        if torch.any(row_indices < 0):
            raise ValueError("row_indices contains negative entries.")
        if torch.any(row_indices >= num_rows):
            raise ValueError("row_indices contains entries >= num_rows.")
    per_row = torch.bincount(row_indices, minlength=num_rows)
    crow = torch.zeros(num_rows + 1, dtype=row_indices.dtype, device=row_indices.device)
    crow[1:] = torch.cumsum(per_row, dim=0)
    return crow


def convert_coo_to_csr_indices_values(
    coo_indices: torch.Tensor, num_rows: int, values: Optional[torch.Tensor] = None
):
    """COO indices (2 or 3 rows) -> (crow, col, values-or-permutation).

    Mirrors reference ``utils/utils.py:236-346`` including its messages; the batched branch builds
    all row pointers with one ``bincount`` instead of a Python loop over batch items.
    """
    n_idx_rows = coo_indices.shape[0]
    if n_idx_rows < 2:
        raise ValueError(
            f"Indices tensor must have at least 2 rows (row and column indices). Got {coo_indices.shape[0]} rows."
        )
    elif n_idx_rows > 3:
        raise ValueError(
            f"Current implementation only supports single batch diomension, therefore indices tensor must have at most 3 rows (batch, row and column indices). Got {coo_indices.shape[0]} rows."
        )
    if coo_indices[-2].max() >= num_rows:
        raise ValueError(
            f"Row indices must be less than num_rows ({num_rows}). Got max row index {coo_indices[-2].max()}"
        )
    if values is not None and values.shape[0] != coo_indices.shape[1]:
        raise ValueError(
            f"Number of values ({values.shape[0]}) does not match number of indices ({coo_indices.shape[1]})"
        )

    sorted_idx, perm = _sort_coo_indices(coo_indices)
    payload = perm if values is None else values[perm]

    if n_idx_rows == 2:
        rows, cols = sorted_idx
        return _compress_row_indices(rows, num_rows), cols, payload

    batch_idx, rows, cols = sorted_idx
    batch_ids = torch.unique(batch_idx)
    n_batch = batch_ids.shape[0]
    # rank of each entry's batch id among the batch ids present (ids need not be 0..b-1)
    rank = torch.searchsorted(batch_ids, batch_idx)
    per_row = torch.bincount(rank * num_rows + rows, minlength=n_batch * num_rows).view(n_batch, num_rows)
    crow = torch.zeros((n_batch, num_rows + 1), dtype=rows.dtype, device=rows.device)
    crow[:, 1:] = torch.cumsum(per_row, dim=1)
    return crow, cols.reshape(n_batch, -1), payload.reshape(n_batch, -1)


def convert_coo_to_csr(sparse_coo_tensor: torch.Tensor) -> torch.Tensor:
    """COO tensor (2-D or one batch dim) -> CSR tensor.  Mirrors reference ``utils/utils.py:349-410``."""
    if sparse_coo_tensor.layout != torch.sparse_coo:
        raise ValueError(f"Unsupported layout: {sparse_coo_tensor.layout}")
    t = sparse_coo_tensor if sparse_coo_tensor.is_coalesced() else sparse_coo_tensor.coalesce()
    crow, col, val = convert_coo_to_csr_indices_values(t.indices(), t.size()[-2], t.values())
    return torch.sparse_csr_tensor(crow, col, val, t.size())


def _demcompress_crow_indices(crow_indices: torch.Tensor, num_rows: int) -> torch.Tensor:
    """CSR row pointer -> one row index per stored entry.  Mirrors reference ``utils/utils.py:413-470``."""
    ids = torch.arange(num_rows, dtype=crow_indices.dtype, device=crow_indices.device)
    return torch.repeat_interleave(ids, crow_indices[1:] - crow_indices[:-1])


def sparse_block_diag(*sparse_tensors: torch.Tensor) -> torch.Tensor:
    """Block-diagonal concatenation of 2-D COO or CSR tensors.

    Mirrors reference ``utils/utils.py:474-645``.  Offsets are computed from the static shapes on
    the host and the nnz prefix from tensor sizes, so assembling the result issues no device→host
    sync (the reference clones ``crow[-1]`` per block).
    """
    for i, t in enumerate(sparse_tensors):
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"TypeError: expected Tensor as element {i} in argument 0, but got {type(t).__name__}")
    if len(sparse_tensors) == 0:
        raise ValueError("At least one sparse tensor must be provided.")
    layouts = {t.layout for t in sparse_tensors}
    if layouts == {torch.sparse_coo}:
        layout = torch.sparse_coo
    elif layouts == {torch.sparse_csr}:
        layout = torch.sparse_csr
    else:
        raise ValueError("Sparse tensors must either be all sparse_coo or all sparse_csr.")
    if any(t.sparse_dim() != 2 for t in sparse_tensors):
        raise ValueError("All sparse tensors must have exactly two sparse dimensions.")
    if any(t.dense_dim() != 0 for t in sparse_tensors):
        raise ValueError("All sparse tensors must have zero dense dimensions.")
    if len(sparse_tensors) == 1:
        return sparse_tensors[0]

    n_tot = sum(t.size(-2) for t in sparse_tensors)
    m_tot = sum(t.size(-1) for t in sparse_tensors)

    if layout == torch.sparse_coo:
        idx_parts, val_parts = [], []
        r_off = c_off = 0
        for t in sparse_tensors:
            t = t if t.is_coalesced() else t.coalesce()
            idx = t.indices()
            shift = torch.tensor([[r_off], [c_off]], dtype=idx.dtype, device=idx.device)
            idx_parts.append(idx + shift)
            val_parts.append(t.values())
            r_off += t.size(-2)
            c_off += t.size(-1)
        return torch.sparse_coo_tensor(torch.cat(idx_parts, dim=1), torch.cat(val_parts), size=(n_tot, m_tot))

    crow_parts, col_parts, val_parts = [], [], []
    nnz_off = c_off = 0
    for k, t in enumerate(sparse_tensors):
        crow, col, val = t.crow_indices(), t.col_indices(), t.values()
        crow_parts.append(crow if k == 0 else crow[1:] + nnz_off)
        col_parts.append(col + c_off)
        val_parts.append(val)
        nnz_off += col.shape[0]
        c_off += t.size(-1)
    return torch.sparse_csr_tensor(
        torch.cat(crow_parts), torch.cat(col_parts), torch.cat(val_parts), size=(n_tot, m_tot)
    )


def sparse_block_diag_split(sparse_block_diag_tensor: torch.Tensor, *shapes: Tuple[int, int]) -> Tuple[torch.Tensor, ...]:
    """Inverse of :func:`sparse_block_diag`.  Mirrors reference ``utils/utils.py:648-790``.

    The CSR branch fetches all block boundaries of the row pointer with ONE device→host copy
    (the reference does two ``.item()`` syncs per block, ``utils/utils.py:766-767``).
    """
    t = sparse_block_diag_tensor
    if not isinstance(t, torch.Tensor):
        raise TypeError("Input must be a torch.Tensor.")
    if t.layout not in (torch.sparse_coo, torch.sparse_csr):
        raise ValueError("Input tensor layout not supported. Only sparse_coo and sparse_csr are supported.")
    if not all(len(s) == 2 for s in shapes):
        raise ValueError("All shapes must be two-dimensional (rows, cols).")
    n_tot = sum(s[0] for s in shapes)
    m_tot = sum(s[1] for s in shapes)
    if (n_tot, m_tot) != (t.size(-2), t.size(-1)):
        raise ValueError(
            f"Sum of provided block shapes ({n_tot}, {m_tot}) does not match "
            f"input tensor size ({t.size(-2)}, {t.size(-1)})."
        )

    blocks = []
    if t.layout == torch.sparse_coo:
        t = t if t.is_coalesced() else t.coalesce()
        rows, cols = t.indices()
        vals = t.values()
        r_off = c_off = 0
        for n, m in shapes:
            keep = (rows >= r_off) & (rows < r_off + n) & (cols >= c_off) & (cols < c_off + m)
            sub = torch.stack((rows[keep] - r_off, cols[keep] - c_off), dim=0)
            blocks.append(torch.sparse_coo_tensor(sub, vals[keep], size=(n, m), device=t.device, dtype=vals.dtype))
            r_off += n
            c_off += m
        return tuple(blocks)

    crow, col, vals = t.crow_indices(), t.col_indices(), t.values()
    bounds = [0]
    for n, _ in shapes:
        bounds.append(bounds[-1] + n)
    ptr_at = crow[torch.tensor(bounds, device=crow.device)].tolist()
    c_off = 0
    for k, (n, m) in enumerate(shapes):
        lo, hi = ptr_at[k], ptr_at[k + 1]
        sub_crow = crow[bounds[k] : bounds[k + 1] + 1] - lo
        blocks.append(
            torch.sparse_csr_tensor(sub_crow, col[lo:hi] - c_off, vals[lo:hi], size=(n, m), device=t.device, dtype=vals.dtype)
        )
        c_off += m
    return tuple(blocks)


def sparse_eye(
    size: Sequence[int],
    *,
    layout: torch.layout = torch.sparse_coo,
    values_dtype: torch.dtype = torch.float64,
    indices_dtype: torch.dtype = torch.int64,
    device: torch.device = torch.device("cpu"),
    requires_grad: bool = False,
) -> torch.Tensor:
    """(Batched) sparse identity.  Mirrors reference ``utils/utils.py:793-912``."""
    if len(size) < 2:
        raise ValueError("size must have at least 2 dimensions")
    if len(size) > 3:
        raise ValueError("size must have at most 3 dimensions (supports 1 batch dimension)")
    if size[-2] != size[-1]:
        raise ValueError("size must define a square matrix (n, n) or batched square matrix (b, n, n)")
    if values_dtype not in (torch.float32, torch.float64):
        raise ValueError(f"Values dtype {values_dtype} not supported. Use torch.float32 or torch.float64.")
    if indices_dtype not in _INT:
        raise ValueError(f"indices_dtype {indices_dtype} not supported. Use torch.int32 or torch.int64.")
    n = size[-1]
    b = size[0] if len(size) == 3 else None
    diag = torch.arange(n, dtype=indices_dtype, device=device)
    ones = torch.ones(n, dtype=values_dtype, device=device)
    if layout == torch.sparse_coo:
        if b is None:
            idx, vals = torch.stack([diag, diag]), ones
        else:
            batch = torch.arange(b, dtype=indices_dtype, device=device).repeat_interleave(n)
            idx = torch.stack([batch, diag.repeat(b), diag.repeat(b)])
            vals = ones.repeat(b)
        return torch.sparse_coo_tensor(
            idx, vals, size, dtype=values_dtype, device=device, requires_grad=requires_grad, is_coalesced=True
        )
    if layout == torch.sparse_csr:
        crow = torch.arange(n + 1, dtype=indices_dtype, device=device)
        col, vals = diag, ones
        if b is not None:
            crow, col, vals = crow.repeat(b, 1), col.repeat(b, 1), vals.repeat(b, 1)
        return torch.sparse_csr_tensor(
            crow, col, vals, size, dtype=values_dtype, device=device, requires_grad=requires_grad
        )
    raise ValueError("Layout {} not supported. Only sparse_coo and sparse_csr are supported.".format(layout))
