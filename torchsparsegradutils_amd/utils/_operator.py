"""Matrix–vector closures for the Krylov loops.

The reference passes ``A.matmul`` of a torch sparse tensor into its solvers
(``utils/linear_cg.py:243-244``, ``utils/bicgstab.py:129-130``, ``utils/minres.py:149-150``), i.e.
one ATen sparse addmm per iteration.  ``as_operator`` turns a sparse tensor into a closure over
the K1 HIP kernel (optionally with the fused pᵀ(Ap) epilogue CG needs)."""

from __future__ import annotations

import torch

from .. import _backend as _be
from .. import _ops
from .. import _pattern as _pt


class SparseOperator:
    """y = A·v through ``tsgu_csr_spmm`` for a 2-D sparse COO/CSR tensor on the GPU."""

    def __init__(self, A: torch.Tensor):
        if A.dim() != 2:
            raise RuntimeError("sparse operator must be a 2-D sparse tensor")
        if A.layout == torch.sparse_csr:
            self.plan = _pt.from_csr(A)
            self.values = A.values()
        elif A.layout == torch.sparse_coo:
            A = A if A.is_coalesced() else A.coalesce()
            self.plan = _pt.from_coo_2d(A.indices(), A.shape, coalesced=True)
            self.values = A.values()
        else:
            raise RuntimeError(f"unsupported sparse layout {A.layout}")
        self.shape = A.shape
        self.dtype = self.values.dtype

    def _cast(self, v):
        # mixed dtypes: the reference's ``A.matmul(v)`` raises torch's dtype error (sparse_generic_solve only warns
        # beforehand, sparse_solve.py:398-403); the raw-pointer kernels must never see operands of two widths
        if v.dtype != self.dtype:
            names = {torch.float64: "Double", torch.float32: "Float", torch.bfloat16: "BFloat16", torch.float16: "Half"}
            raise RuntimeError(f"expected scalar type {names.get(self.dtype, self.dtype)} but found {names.get(v.dtype, v.dtype)}")
        return v

    def __call__(self, v: torch.Tensor) -> torch.Tensor:
        v = self._cast(v)
        p = self.plan
        if v.dim() == 1:
            return _ops.spmm(p, self.values, v.unsqueeze(-1)).squeeze(-1)
        return _ops.spmm(p, self.values, v)

    matmul = __call__

    def matmul_with_dot(self, v: torch.Tensor, w: torch.Tensor = None, out: torch.Tensor = None, skip: int = 0):
        """(A·v, per-block partial sums of <w, A·v> per column; w defaults to v) — K1 with the fused dot epilogue.
        ``out`` (optional, must not alias ``v``) receives A·v in place.  ``skip``: address of a device int32; kernels that can
        (the plane sweep) do nothing when it is non-zero — a hint, the result is then unspecified."""
        p = self.plan
        v = self._cast(v)
        if v.dtype in (torch.float32, torch.float64) and self.values.dtype == v.dtype and v.dim() == 2:
            # stencil on a lattice: the plane sweep needs no column indices, and the own row of v is already in LDS for the dot (a second
            # operand w is read from memory; `out` receives the product in place)
            plain = ((w is None or (w.shape == v.shape and w.dtype == v.dtype and w.is_contiguous() and w.data_ptr() % 16 == 0))
                     and (out is None or (out.shape == (p.n_rows, v.size(-1)) and out.dtype == v.dtype and out.is_contiguous()
                                          and out.data_ptr() % 16 == 0 and out.data_ptr() != v.data_ptr())))
            got = _ops._lattice_cfg(p, _be.LAT_SPMM, v) if plain and p.perm is None and p.batch is None else None
            if got is not None and not getattr(got[1], "march", False) and got[1].cpl == 1:
                return _be.csr_spmm_lattice(got[0], got[1], self.values, v, dot=True, skip=skip, dot_w=w, out=out)
        return _be.csr_spmm(p.crow, p.col, self.values, v, p.n_rows, p.n_cols, perm=p.perm, out=out,
                            dot_w=v if w is None else w, max_row_nnz=p.max_row_nnz)


def checked(out: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Output of a matrix–vector closure inside a fused solver loop: the raw-pointer kernels work in ONE dtype, so a
    closure that promotes (e.g. a float64 operator applied to float32 vectors) is an error, as it is — with torch's
    own message — in the reference's op chain."""
    if out.dtype != dtype:
        raise RuntimeError(f"matmul_closure returned {out.dtype} for {dtype} vectors: expected one working dtype")
    return out


def as_operator(matmul_closure):
    """tensor-or-callable → callable, keeping the reference's error for anything else."""
    if torch.is_tensor(matmul_closure):
        if matmul_closure.layout in (torch.sparse_csr, torch.sparse_coo):
            return SparseOperator(matmul_closure)
        return matmul_closure.matmul
    if callable(matmul_closure):
        return matmul_closure
    raise RuntimeError("matmul_closure must be a tensor, or a callable object!")
