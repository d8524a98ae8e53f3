"""``bicgstab`` — drop-in for reference ``torchsparsegradutils/utils/bicgstab.py`` (pykrylov port).

Same signature, settings tuple, defaults and quirks: several right-hand sides are solved column
by column (:113-124); ``matvec_max`` defaults to ``2n`` (:155); with ``initial_guess=None`` one
matvec is spent on ``A·0`` (:159-161) and with an initial guess the residual is *not* corrected
by ``A·x0`` (:158-161); ``rho_next = -omega·<r0, t>`` (:224).  The two matvecs per iteration run
on the K1 HIP SpMM when ``matmul_closure`` is a sparse tensor and every inner product on the
deterministic two-stage HIP reduction; all scalars stay 0-dim device tensors, the host reads only
the two residual-norm comparisons per iteration that steer the loop.
"""

from __future__ import annotations

import logging
from typing import Callable, NamedTuple, Optional, Union

import torch

from .. import _backend as _be
from ._operator import as_operator

_null_log = logging.getLogger("bicgstab")
_null_log.disabled = True


class BICGSTABSettings(NamedTuple):
    """Mirrors reference ``utils/bicgstab.py:14-19``."""

    matvec_max: Optional[int] = None
    abstol: float = 1.0e-8
    reltol: float = 1.0e-6
    precon: Optional[Union[torch.Tensor, Callable[[torch.Tensor], torch.Tensor]]] = None
    logger: logging.Logger = _null_log


def _dot(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """<a, b> of two length-n device vectors as a 0-dim tensor (HIP reduction)."""
    return _be.coldot(a.unsqueeze(-1), b.unsqueeze(-1)).squeeze(0)


def bicgstab(
    matmul_closure: Union[torch.Tensor, Callable[[torch.Tensor], torch.Tensor]],
    rhs: torch.Tensor,
    initial_guess: Optional[torch.Tensor] = None,
    settings: BICGSTABSettings = BICGSTABSettings(),
) -> torch.Tensor:
    r"""Solve :math:`A x = b` for general (non-symmetric) ``A`` with BiCGSTAB.

    ``matmul_closure``: tensor (dense or sparse COO/CSR) or callable; ``rhs``: ``(n,)`` or ``(n, k)``
    on the GPU.  Returns the solution with the shape of ``rhs``."""
    _be.require_device(rhs)
    if rhs.dim() > 1:
        # column-by-column, exactly like the reference (each column has its own stopping point)
        sols = [
            bicgstab(matmul_closure, rhs[:, i], None if initial_guess is None else initial_guess[:, i], settings)
            for i in range(rhs.shape[1])
        ]
        return torch.stack(sols, dim=1)

    n = rhs.shape[0]
    op = as_operator(matmul_closure)

    if settings.precon is None:
        precon = None
    elif torch.is_tensor(settings.precon):
        precon = settings.precon.matmul
    elif callable(settings.precon):
        precon = settings.precon
    else:
        raise RuntimeError("settings.precon must be a tensor, or a callable object!")

    rhs = rhs.contiguous()
    x = torch.zeros(n, dtype=rhs.dtype, device=rhs.device) if initial_guess is None else initial_guess.clone()
    matvec_max = 2 * n if settings.matvec_max is None else settings.matvec_max
    n_matvec = 0

    r0 = rhs.clone()
    if initial_guess is None:
        r0 = rhs - op(x)
        n_matvec += 1

    rho = alpha = omega = 1.0
    rho_next = _dot(r0, r0)
    resid = resid0 = torch.abs(torch.sqrt(rho_next))
    threshold = max(settings.abstol, settings.reltol * float(resid0))
    finished = bool(resid <= threshold) or n_matvec >= matvec_max

    log = settings.logger
    log.info("Initial residual = %8.2e" % float(resid0))
    log.info("Threshold = %8.2e" % threshold)

    if not finished:
        r = r0.clone()
        p = torch.zeros_like(rhs)
        v = torch.zeros_like(rhs)

    while not finished:
        beta = rho_next / rho * alpha / omega
        rho = rho_next
        # p = r + beta·(p − omega·v)   (reference :186-191)
        p = torch.addcmul(r, beta, p - omega * v)
        q = precon(p) if precon is not None else p

        v = op(q)
        n_matvec += 1
        alpha = rho / _dot(r0, v)
        s = r - alpha * v
        resid = torch.sqrt(_dot(s, s))
        log.info("%6d  %8.2e" % (n_matvec, float(resid)) if not log.disabled else "")

        if bool(resid <= threshold):
            x = x + alpha * q
            break
        if n_matvec >= matvec_max:
            break

        z = precon(s) if precon is not None else s
        t = op(z)
        n_matvec += 1
        omega = _dot(t, s) / _dot(t, t)
        rho_next = -omega * _dot(r0, t)
        r = s - omega * t
        x = x + omega * z + alpha * q
        resid = torch.sqrt(_dot(r, r))
        log.info("%6d  %8.2e" % (n_matvec, float(resid)) if not log.disabled else "")
        if bool(resid <= threshold) or n_matvec >= matvec_max:
            break

    return x
