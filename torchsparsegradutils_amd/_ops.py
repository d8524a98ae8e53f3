"""Kernel selection for one sparse operand: LDS-tiled kernels when the pattern has column reuse
inside row blocks (stencils, banded factors) and the operands qualify, plain gather kernels otherwise.
Both produce the same values (same summation order); the choice is speed only."""

from __future__ import annotations

import os

import torch

from . import _backend as _be
from ._pattern import RowGather

# The wave-pipelined LDS-tiled kernels are parity-clean but only on par with / slightly ahead of the gather
# kernels: off by default; TSGU_ENABLE_TILED=1 selects them where the pattern qualifies.
ENABLE_TILED = os.environ.get("TSGU_ENABLE_TILED", "0") == "1"


# Block-dictionary kernels (csrc/blocktile_impl.h, gather-from-global flavour): used for operands addressed through
# a permutation (transposed patterns: gradB = Aᵀ·G and the fused backward), where the per-block sorted permutation
# turns the 4-byte scattered value / gradA accesses into short runs.  TSGU_ENABLE_BLOCK=0 falls back to K2 / the
# plain fused backward.  Patterns below the size threshold stay on the plan-free kernels (launch-bound anyway).
ENABLE_BLOCK = os.environ.get("TSGU_ENABLE_BLOCK", "1") == "1"
BLOCK_MIN_NNZ = 1 << 16


def _block_for(plan: RowGather, dense: torch.Tensor, *others: torch.Tensor):
    if (not ENABLE_BLOCK or plan.batch is not None or plan.perm is None or dense.dim() != 2
            or plan.nnz < BLOCK_MIN_NNZ or plan.crow.dtype not in (torch.int32, torch.int64)):
        return None
    geo = _be.blocktile_limits(dense.dtype, dense.size(-1), tile=False)
    if geo is None or not _be._tiled_ok(*(_be.rowmajor(t) for t in (dense,) + others)):
        return None
    return plan.block_plan(*geo)


# Row-pair kernels (csrc/rowpack_impl.h): rows 2q, 2q+1 walk the union of their columns, a shared dense row is
# gathered once.  First choice for forward, transposed and fused-backward walks when neighbouring rows share columns
# (stencil / banded / mesh patterns: C2 K1 151 -> 131 us, fused backward 412 -> 312 us); TSGU_ENABLE_PACK=0 disables.
ENABLE_PACK = os.environ.get("TSGU_ENABLE_PACK", "1") == "1"
PACK_MIN_NNZ = 1 << 16


def _pack_for(plan: RowGather, dense: torch.Tensor, *others: torch.Tensor):
    if (not ENABLE_PACK or plan.batch is not None or dense.dim() != 2 or plan.nnz < PACK_MIN_NNZ
            or plan.crow.dtype not in (torch.int32, torch.int64)):
        return None
    geo = _be.rowpack_limits(dense.dtype, dense.size(-1))
    if geo is None or not _be._tiled_ok(*(_be.rowmajor(t) for t in (dense,) + others)):
        return None
    return plan.rowpack_plan(*geo)


def mm_backward(plan: RowGather, values: torch.Tensor, G: torch.Tensor, B: torch.Tensor):
    """(gradA values in A's order, gradB) of C = A·B in one pass over the transposed pattern."""
    t = plan.transposed
    same = values.dtype == G.dtype == B.dtype
    rp = _pack_for(t, G, B) if same else None
    if rp is not None:
        return _be.csr_mm_backward_rowpack(t.crow, rp, values, G, B, t.n_rows)
    bp = _block_for(t, G, B) if same else None
    if bp is not None:
        return _be.csr_mm_backward_blocktile(t.crow, bp, values, G, B, t.n_rows, tile=False)
    return _be.csr_mm_backward(t, values, G, B, plan.n_rows, plan.n_cols)


def _tiles_for(plan: RowGather, dense: torch.Tensor, *others: torch.Tensor):
    if not ENABLE_TILED or plan.batch is not None or dense.dim() != 2:
        return None
    geo = _be.tiled_geometry(dense.dtype, dense.size(-1))
    if geo is None:
        return None
    if not _be._tiled_ok(*(_be.rowmajor(t) for t in (dense,) + others)):
        return None
    return plan.tiles(*geo)


def spmm(plan: RowGather, values: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A·B for the operand described by (plan, values); perm-aware (transposed / un-coalesced plans)."""
    rp = _pack_for(plan, B) if values.dtype == B.dtype else None
    if rp is not None:
        return _be.csr_spmm_rowpack(plan.crow, values, rp, B, plan.n_rows)
    bp = _block_for(plan, B) if values.dtype == B.dtype else None
    if bp is not None:
        return _be.csr_spmm_blocktile(plan.crow, values, bp, B, plan.n_rows, tile=False)
    tiles = _tiles_for(plan, B) if values.dtype == B.dtype else None
    if tiles is not None:
        return _be.csr_spmm_tiled(plan.crow, values, tiles, B, plan.n_rows, plan.n_cols, perm=plan.perm)
    return _be.csr_spmm(plan.crow, plan.col, values, B, plan.n_rows, plan.n_cols, perm=plan.perm)


def sddmm(plan: RowGather, G: torch.Tensor, B: torch.Tensor, alpha: float = 1.0, swap_roles: bool = False) -> torch.Tensor:
    """alpha·<G[row k], B[col k]> (or roles swapped) at the plan's stored entries, in plan order."""
    gathered = G if swap_roles else B
    rowop = B if swap_roles else G
    if plan.perm is None and G.dtype == B.dtype:
        rp = _pack_for(plan, gathered, rowop)
        if rp is not None:
            return _be.csr_sddmm_rowpack(plan.crow, rp, rowop, gathered, plan.n_rows, alpha=alpha)
    tiles = _tiles_for(plan, gathered, B if swap_roles else G) if G.dtype == B.dtype else None
    if tiles is not None:
        return _be.csr_sddmm_tiled(plan.crow, tiles, G, B, plan.n_rows, plan.n_cols, alpha=alpha, swap_roles=swap_roles)
    return _be.csr_sddmm(plan.crow, plan.col, G, B, plan.n_rows, plan.n_cols, alpha=alpha, swap_roles=swap_roles)
