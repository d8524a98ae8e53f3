"""``minres`` — drop-in for reference ``torchsparsegradutils/utils/minres.py`` (the default solver of
``sparse_generic_solve``, reference sparse_solve.py:406-410).

MINRES is not a kernel target of this build: the Lanczos + Givens
recurrences below are the same mathematics as the reference, expressed as device tensor ops; what
runs on the hand-written HIP path is the matvec (K1 SpMM when ``matmul_closure`` is a sparse
tensor).  Signature, settings, the rhs normalisation, the ``max_iter = min(max_iter, n+1)`` cap,
the every-10-iterations relative-update stopping test and the shifted-system output layout follow
reference ``utils/minres.py:140-311``.
"""

from __future__ import annotations

from typing import Callable, NamedTuple, Optional, Union

import torch

from .. import _backend as _be
from ._operator import as_operator


class MINRESSettings(NamedTuple):
    """Mirrors reference ``utils/minres.py:9-13``."""

    max_cg_iterations: int = 1000
    minres_tolerance: float = 1e-4
    verbose_linalg: bool = False


def minres(
    matmul_closure: Union[torch.Tensor, Callable[[torch.Tensor], torch.Tensor]],
    rhs: torch.Tensor,
    eps: float = 1e-25,
    shifts: Optional[torch.Tensor] = None,
    value: Optional[float] = None,
    max_iter: Optional[int] = None,
    preconditioner: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
    settings: MINRESSettings = MINRESSettings(),
) -> torch.Tensor:
    r"""Solve symmetric (possibly indefinite) systems :math:`(A + \sigma I) x = b` with MINRES."""
    _be.require_device(rhs)
    mm = as_operator(matmul_closure)
    precond = (lambda v: v.clone()) if preconditioner is None else preconditioner

    if shifts is None:
        shifts = torch.tensor(0.0, dtype=rhs.dtype, device=rhs.device)
    squeeze = rhs.dim() == 1
    if squeeze:
        rhs = rhs.unsqueeze(-1)

    rhs_norm = torch.linalg.vector_norm(rhs, ord=2, dim=-2, keepdim=True)
    rhs_is_zero = rhs_norm.lt(1e-10)
    rhs_norm = rhs_norm.masked_fill_(rhs_is_zero, 1)
    rhs = rhs.div(rhs_norm)

    if max_iter is None:
        max_iter = settings.max_cg_iterations
    max_iter = min(max_iter, rhs.size(-2) + 1)
    eps_t = torch.tensor(eps, dtype=rhs.dtype, device=rhs.device)

    def apply(v):
        out = mm(v)
        return out.mul(value) if value is not None else out

    probe = apply(rhs)
    shifts = shifts.reshape(shifts.shape + (1,) * (probe.dim() - shifts.dim() + 1))
    n_shift = shifts.shape[0]
    sol = torch.zeros((n_shift,) + tuple(probe.shape), dtype=rhs.dtype, device=rhs.device)

    # Lanczos state
    z_pp = torch.zeros_like(probe)
    z_p = rhs.clone().expand_as(probe).contiguous()
    q_p = precond(z_p)
    beta_p = (z_p * q_p).sum(dim=-2, keepdim=True).sqrt()
    z_p = z_p / beta_p
    q_p = q_p / beta_p

    # Givens state (one set per shift)
    shape_s = tuple(sol.shape[:-2]) + (1, rhs.size(-1))
    c_pp = torch.ones(shape_s, dtype=rhs.dtype, device=rhs.device)
    s_pp = torch.zeros_like(c_pp)
    c_p = torch.ones_like(c_pp)
    s_p = torch.zeros_like(c_pp)
    w_pp = torch.zeros_like(sol)
    w_p = torch.zeros_like(sol)
    scale_p = beta_p.repeat(n_shift, *([1] * beta_p.dim()))
    update = torch.zeros_like(sol)

    if settings.verbose_linalg:
        print(
            f"Running MINRES on a {rhs.shape} RHS for {max_iter} iterations (tol={settings.minres_tolerance}). "
            f"Output: {sol.shape}."
        )

    for i in range(max_iter + 2):
        prod = apply(q_p)
        alpha = (prod * q_p).sum(dim=-2, keepdim=True)
        z_c = prod - alpha * z_p - beta_p * z_pp
        q_c = precond(z_c)
        beta_c = (z_c * q_c).sum(dim=-2, keepdim=True).sqrt().clamp_min(eps_t)
        z_c = z_c / beta_c
        q_c = q_c / beta_c

        # QR of the shifted tridiagonal by Givens rotations
        subsub = s_pp * beta_p
        sub = c_pp * beta_p
        alpha_s = alpha + shifts
        diag = alpha_s * c_p - s_p * sub
        sub = sub * c_p + s_p * alpha_s
        radius = (diag * diag + beta_c * beta_c).sqrt()
        c_c = diag / radius
        s_c = beta_c / radius
        diag = diag * c_c + s_c * beta_c

        scale_c = -(scale_p * s_c)
        scale_p = scale_p * c_c
        w_c = (q_p - sub * w_p - subsub * w_pp) / diag
        update = w_c * scale_p
        sol = sol + update

        if (i + 1) % 10 == 0:
            un = torch.linalg.vector_norm(update, dim=-2)
            sn = torch.linalg.vector_norm(sol, dim=-2)
            if (un / sn).mean().item() < settings.minres_tolerance:
                break

        z_pp, z_p = z_p, z_c
        q_p = q_c
        beta_p = beta_c
        c_pp, c_p = c_p, c_c
        s_pp, s_p = s_p, s_c
        w_pp, w_p = w_p, w_c
        scale_p = scale_c

    sol = sol.masked_fill(rhs_is_zero, 0)
    if squeeze:
        sol = sol.squeeze(-1)
        rhs_norm = rhs_norm.squeeze(-1)
    if shifts.numel() == 1:
        sol = sol.squeeze(0)
    return sol.mul(rhs_norm)
