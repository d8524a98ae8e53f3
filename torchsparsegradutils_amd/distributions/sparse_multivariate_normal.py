"""``SparseMultivariateNormal`` — the real caller of the hot path (reference
``torchsparsegradutils/distributions/sparse_multivariate_normal.py:105-389``): a multivariate normal whose
covariance or precision is given by a sparse lower-triangular factor (``L Lᵀ`` or ``L D Lᵀ`` with unit ``L``), sampled
with the reparameterisation trick.

``rsample`` is two calls into the hot path:

* covariance factor:  ``x = L·ε``  (``+ η`` for the implicit unit diagonal of the LDLᵀ form)        → K1 ``sparse_mm``
* precision factor:   ``x = L⁻ᵀ·ε`` → K4 ``sparse_triangular_solve(upper=False, transpose=True[, unitriangular=True])``

The reference routes both through ``_batch_sparse_mv``, which hands the operators a TRANSPOSED VIEW of the noise
(``bvec.t()``, :96) and transposes the result back.  Here that view is consumed in place: K1 / K4 take a column stride,
so the sequence allocates the result and nothing else (the reference's backends copy the view to row-major first).
Constructor checks and messages follow the reference (:249-322).  Only what ``rsample`` needs is provided; ``log_prob``
and the ``Native`` variant are outside the hot path.
"""

from __future__ import annotations

import torch
from torch.distributions import constraints
from torch.distributions.distribution import Distribution
from torch.distributions.utils import _standard_normal

from ..sparse_matmul import sparse_mm
from ..sparse_solve import sparse_triangular_solve


def _apply_to_samples(op, mat: torch.Tensor, vec: torch.Tensor, **kwargs) -> torch.Tensor:
    """``op(mat, ·)`` on every sample of ``vec``: samples are rows of ``vec``, the operators want them as columns
    (reference ``_batch_sparse_mv`` :91-102; same four rank combinations, no broadcasting of batch dimensions)."""
    if mat.dim() == 2 and vec.dim() == 1:
        return op(mat, vec.unsqueeze(-1), **kwargs).squeeze(-1)
    if mat.dim() == 2 and vec.dim() == 2:
        return op(mat, vec.t(), **kwargs).t()           # (n, k) transposed view in, transposed layout out: no copies
    if mat.dim() == 3 and vec.dim() == 2:
        return op(mat, vec.unsqueeze(-1), **kwargs).squeeze(-1)
    if mat.dim() == 3 and vec.dim() == 3:
        return op(mat, vec.permute(1, 2, 0), **kwargs).permute(2, 0, 1)
    raise ValueError("Invalid dimensions for bmat and bvec")


_batch_sparse_mv = _apply_to_samples  # the reference's name for the helper


def _factor(name: str, t: torch.Tensor) -> torch.Tensor:
    if t.layout == torch.sparse_coo:
        t = t if t.is_coalesced() else t.coalesce()
    elif t.layout != torch.sparse_csr:
        raise ValueError("{} must be sparse COO or CSR, instead of {}".format(name, t.layout))
    if t.dim() < 2:
        raise ValueError(f"{name} {'matrix ' if name == 'scale_tril' else ''}must be at least two-dimensional, "
                         f"with optional leading batch dimension{'' if name == 'scale_tril' else 's'}")
    if t.dim() > 3:
        raise ValueError("{} can only have 1 batch dimension, but has {}".format(name, t.dim() - 2))
    return t


class SparseMultivariateNormal(Distribution):
    r"""Multivariate normal :math:`\mathcal N(\mu, \Sigma)` with :math:`\Sigma = L L^\top`, :math:`L D L^\top` (``scale_tril``)
    or :math:`\Sigma^{-1} = L L^\top`, :math:`L D L^\top` (``precision_tril``); ``diagonal`` given ⇒ LDLᵀ with unit,
    strictly-lower-stored ``L``.  ``loc``: ``(n,)`` or ``(B, n)``; factors: sparse COO/CSR ``(n, n)`` or ``(B, n, n)``."""

    support = constraints.real_vector
    has_rsample = True
    arg_constraints = {}

    def __init__(self, loc, diagonal=None, scale_tril=None, precision_tril=None, validate_args=None):
        if loc.dim() < 1:
            raise ValueError("loc must be at least one-dimensional.")
        if loc.dim() > 2:
            raise ValueError(
                "loc must be at most two-dimensional as the current implementation only supports 1 batch dimension."
            )
        event_shape = loc.shape[-1:]
        self._loc = loc
        if diagonal is not None:
            if diagonal.dim() < 1:
                raise ValueError("diagonal must be at least one-dimensional.")
            if diagonal.dim() > 2:
                raise ValueError(
                    "diagonal must be at most two-dimensional as the current implementation only supports 1 batch dimension."
                )
            if diagonal.shape[-1:] != event_shape:
                raise ValueError("diagonal must be a batch of vectors with shape {}".format(event_shape))
        self._diagonal = diagonal
        if (scale_tril is not None) + (precision_tril is not None) != 1:
            raise ValueError("Exactly one of scale_tril or precision_tril may be specified.")
        if scale_tril is not None:
            factor = self._scale_tril = _factor("scale_tril", scale_tril)
        else:
            factor = self._precision_tril = _factor("precision_tril", precision_tril)
        shapes = [loc.shape[:-1], factor.shape[:-2]] + ([diagonal.shape[:-1]] if diagonal is not None else [])
        super().__init__(torch.broadcast_shapes(*shapes), event_shape, validate_args=validate_args)

    diagonal = property(lambda self: self._diagonal)
    scale_tril = property(lambda self: self._scale_tril)
    precision_tril = property(lambda self: self._precision_tril)
    loc = property(lambda self: self._loc)
    mean = property(lambda self: self._loc)
    mode = property(lambda self: self._loc)

    @property
    def is_ldlt_parameterization(self):
        return self._diagonal is not None

    def _transform(self, eps: torch.Tensor) -> torch.Tensor:
        """Standard-normal noise → sample (reference :358-389)."""
        ldlt = self._diagonal is not None
        if "_scale_tril" in self.__dict__:
            if ldlt:
                eta = self._diagonal.sqrt() * eps
                x = _apply_to_samples(sparse_mm, self._scale_tril, eta) + eta   # unit diagonal is implicit
            else:
                x = _apply_to_samples(sparse_mm, self._scale_tril, eps)
        else:
            rhs = eps / self._diagonal.sqrt() if ldlt else eps
            x = _apply_to_samples(sparse_triangular_solve, self._precision_tril, rhs,
                                  upper=False, unitriangular=ldlt, transpose=True)
        return self._loc + x

    def rsample(self, sample_shape=torch.Size()):
        shape = self._extended_shape(sample_shape)
        return self._transform(_standard_normal(shape, dtype=self._loc.dtype, device=self._loc.device))
