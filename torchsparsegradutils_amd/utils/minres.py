"""``minres`` — drop-in for reference ``torchsparsegradutils/utils/minres.py`` (the default solver of
``sparse_generic_solve``, reference sparse_solve.py:406-410).

Without a preconditioner and with a single shift (the ``sparse_generic_solve`` default path) the
Lanczos + Givens recurrences run on the fused K7 kernels (``csrc/minres.hip``): five launches per
iteration around the K1 SpMM, all per-column scalars on the device, one host read every 10
iterations — where the reference synchronises for its stopping test — and hipGraph replay of
10-iteration chunks for long solves.  A preconditioner, several shifts or a ``value`` factor use
the same mathematics as device tensor ops around the K1 matvec.  Signature, settings, the rhs
normalisation, the ``max_iter = min(max_iter, n+1)`` cap, the every-10-iterations relative-update
stopping test and the shifted-system output layout follow reference ``utils/minres.py:140-311``.
"""

from __future__ import annotations

import threading
from typing import Callable, NamedTuple, Optional, Union

import torch

from .. import _backend as _be
from . import _graph
from ._operator import SparseOperator, as_operator, checked


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

    if (preconditioner is None and value is None and shifts.numel() == 1 and rhs.dim() == 2
            and rhs.dtype in (torch.float32, torch.float64) and 0 < rhs.size(-1) <= 1024 and rhs.size(-2) > 0):
        sol = _minres_fused(mm, rhs.contiguous(), float(shifts), eps, max_iter, settings).unsqueeze(0)
        sol = sol.masked_fill(rhs_is_zero, 0)
        if squeeze:
            sol = sol.squeeze(-1)
            rhs_norm = rhs_norm.squeeze(-1)
        return sol.squeeze(0).mul(rhs_norm)

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


def _minres_fused(op, rhs: torch.Tensor, shift: float, eps: float, max_iter: int, settings: MINRESSettings) -> torch.Tensor:
    """Un-preconditioned single-shift MINRES on the K7 kernels; `rhs` is the normalised (n, p) right-hand side
    (reference utils/minres.py:241-311; without a preconditioner q == z)."""
    lib = _be.load_library()
    n, p = rhs.shape
    dev, dtype = rhs.device, rhs.dtype
    vt = _be.vtype_of(rhs)
    fused_dot = isinstance(op, SparseOperator) and op.dtype == dtype
    nb = lib.tsgu_cg_num_blocks(vt, n, p)
    if nb < 0:
        raise RuntimeError("minres: more than 1024 simultaneous right-hand sides are not supported")

    z = [torch.zeros_like(rhs), rhs.clone()]                     # [z two steps back, z one step back]
    beta0 = (z[1] * z[1]).sum(dim=-2).sqrt()                    # (minres.py:246-248)
    z[1] = z[1] / beta0
    w = [torch.zeros_like(rhs), torch.zeros_like(rhs)]
    sol = torch.zeros_like(rhs)
    scal = torch.zeros(12 * p, dtype=dtype, device=dev)
    scal[p : 2 * p] = beta0
    scal[11 * p :] = beta0
    scal[6 * p : 7 * p] = beta0                                  # scale_prev (minres.py:255)
    scal[2 * p : 3 * p] = 1
    scal[4 * p : 5 * p] = 1
    flags = torch.zeros(2, dtype=torch.int32, device=dev)
    part = torch.empty((2, nb, p), dtype=dtype, device=dev)
    fold = torch.empty((lib.tsgu_cg_fold_rows(), p), dtype=dtype, device=dev)
    stream = lambda: torch.cuda.current_stream(dev).cuda_stream  # noqa: E731

    def scalar(phase, partial, rows, set_stride=0):
        _be.check(lib.tsgu_minres_scalar(vt, phase, partial.data_ptr(), rows, set_stride, fold.data_ptr(), scal.data_ptr(),
                                         flags.data_ptr(), float(eps), float(settings.minres_tolerance), shift, p,
                                         dev.index, stream()), "tsgu_minres_scalar")

    def iteration(check: bool):
        if fused_dot:
            prod, pzz = op.matmul_with_dot(z[1])                # A z with <z, A z> partials (minres.py:261-262)
        else:
            prod = checked(op(z[1]), dtype).contiguous()
            pzz = _be.coldot(prod, z[1]).unsqueeze(0).contiguous()
        scalar(0, pzz, pzz.shape[0])
        _be.check(lib.tsgu_minres_vector(vt, 0, n, p, z[0].data_ptr(), z[1].data_ptr(), prod.data_ptr(), None, None,
                                         scal.data_ptr(), flags.data_ptr(), part.data_ptr(), 0, 0, dev.index, stream()),
                  "tsgu_minres_vector")
        scalar(1, part[0], nb)
        _be.check(lib.tsgu_minres_vector(vt, 1, n, p, z[0].data_ptr(), z[1].data_ptr(), w[0].data_ptr(), w[1].data_ptr(),
                                         sol.data_ptr(), scal.data_ptr(), flags.data_ptr(), part.data_ptr(), nb * p,
                                         int(check), dev.index, stream()), "tsgu_minres_vector")
        if check:
            scalar(2, part, nb, nb * p)                          # every 10th iteration (minres.py:299-305)
        z.reverse()                                              # the buffer that held z two steps back now holds z_c
        w.reverse()

    def chunk10():
        for k in range(10):
            iteration(k == 9)

    total = max_iter + 2                                         # (minres.py:259)
    i = 0
    graph = None
    try_graph = fused_dot and _graph.enabled()
    with torch.cuda.device(dev):
        while i < total:
            if i % 10 == 0 and total - i >= 10:
                if try_graph and graph is None and i >= 10 and total - i >= _graph.MIN_ITERS:
                    roles = (z[:], w[:])
                    graph = _graph.capture(chunk10, 1)           # buffer roles return after an even number of steps
                    try_graph = graph is not None
                    if graph is None:
                        z[:], w[:] = roles                       # nothing executed: undo the recorded role swaps
                if graph is not None:
                    _graph.replay(graph)
                else:
                    chunk10()
                i += 10
                if bool(flags[0].item()):
                    break
            else:
                iteration((i + 1) % 10 == 0)
                i += 1
                if i % 10 == 0 and bool(flags[0].item()):
                    break
    _INFO.last = {"solver": "minres", "iterations": i, "tolerance_reached": bool(flags[0].item()),
                  "tolerance": float(settings.minres_tolerance)}
    return sol


class _Info(threading.local):
    last = None


_INFO = _Info()


def last_solve_info():
    """Diagnostics of this thread's most recent fused ``minres`` call (iterations, stopping test outcome)."""
    return _INFO.last
