"""``minres`` — drop-in for reference ``torchsparsegradutils/utils/minres.py`` (the default solver of
``sparse_generic_solve``, reference sparse_solve.py:406-410).

The Lanczos + Givens recurrences run on the fused K7 kernels (``csrc/minres.hip``): five launches
per iteration around the K1 SpMM, all per-column (and per-shift) scalars on the device, one host
read every 10 iterations — where the reference synchronises for its stopping test — and hipGraph
replay of 10-iteration chunks for long solves.  Several shifts share the Lanczos vectors and update
their solutions in one launch; ``value`` scales the product inside the Lanczos kernel; a
preconditioner is called between two of the kernels.  Batched right-hand sides (more than two
dimensions) use the same mathematics as device tensor ops around the K1 matvec.  Signature, settings, the rhs
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


ENABLE_FUSED = True  # False: the reference's op chain around the K1 matvec (tests compare the two)


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
    if (ENABLE_FUSED and rhs.is_cuda and rhs.dim() > 2 and (shifts is None or shifts.dim() <= 1) and rhs.dtype in (torch.float32, torch.float64)
            and rhs.numel() > 0):
        # right-hand sides with batch dimensions (*batch, n, k) (reference utils/minres.py:221-233: norms per (batch, column), the
        # stop rule is their mean): the batch is folded into the columns, (n, batch·k), and solved by the 2-D path on the fused
        # kernels.  A 2-D sparse operator applies to every column alike; any other closure (and a preconditioner) sees its own
        # (*batch, n, k) layout through a reshaping wrapper.
        batch_shape = tuple(rhs.shape[:-2])
        n, k = rhs.shape[-2:]
        nb = rhs.numel() // (n * k)

        def fold(t):      # (*batch, n, k) -> (n, batch*k)
            return t.reshape(nb, n, k).permute(1, 0, 2).reshape(n, nb * k)

        def unfold(t):    # (..., n, batch*k) -> (..., *batch, n, k)
            lead = tuple(t.shape[:-2])
            return t.reshape(lead + (n, nb, k)).movedim(-2, -3).reshape(lead + batch_shape + (n, k))

        if torch.is_tensor(matmul_closure) and matmul_closure.dim() == 2 and matmul_closure.layout in (torch.sparse_csr, torch.sparse_coo):
            op2 = matmul_closure
        else:
            inner = as_operator(matmul_closure)
            op2 = lambda v: fold(inner(unfold(v)))  # noqa: E731
        pre = None if preconditioner is None else (lambda v: fold(preconditioner(unfold(v))))
        return unfold(minres(op2, fold(rhs).contiguous(), eps, shifts, value, max_iter, pre, settings))
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

    n_sh = shifts.numel()
    if (ENABLE_FUSED and rhs.is_cuda and rhs.dim() == 2 and shifts.dim() <= 1 and 0 < n_sh <= 64 and shifts.device == rhs.device
            and rhs.dtype in (torch.float32, torch.float64) and 0 < rhs.size(-1) <= 1024 and rhs.size(-2) > 0):
        sol = _minres_fused(mm, rhs.contiguous(), shifts.reshape(-1).to(rhs.dtype).contiguous(), value, preconditioner, eps,
                            max_iter, settings)
        sol = sol.masked_fill(rhs_is_zero, 0)
        if squeeze:
            sol = sol.squeeze(-1)
            rhs_norm = rhs_norm.squeeze(-1)
        if n_sh == 1:
            sol = sol.squeeze(0)
        return sol.mul(rhs_norm)

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


def _minres_fused(op, rhs: torch.Tensor, shifts: torch.Tensor, value, precond, eps: float, max_iter: int,
                  settings: MINRESSettings) -> torch.Tensor:
    """MINRES on the K7 kernels; `rhs` is the normalised (n, p) right-hand side, `shifts` the (S,) shifts; returns the
    (S, n, p) solutions (reference utils/minres.py:241-311; without a preconditioner q == z)."""
    lib = _be.load_library()
    n, p = rhs.shape
    S = shifts.numel()
    dev, dtype = rhs.device, rhs.dtype
    vt = _be.vtype_of(rhs)
    fused_dot = isinstance(op, SparseOperator) and op.dtype == dtype
    val = 1.0 if value is None else float(value)
    nb = lib.tsgu_cg_num_blocks(vt, n, p)
    if nb < 0:
        raise RuntimeError("minres: more than 1024 simultaneous right-hand sides are not supported")

    z = [torch.zeros_like(rhs), rhs.clone()]                     # [z two steps back, z one step back]
    if precond is None:
        q = None
        beta0 = (z[1] * z[1]).sum(dim=-2).sqrt()                # (minres.py:246-248)
        z[1] = z[1] / beta0
    else:
        q = [None, checked(precond(z[1]), dtype).contiguous()]  # [scratch, q one step back]  (minres.py:245)
        beta0 = (z[1] * q[1]).sum(dim=-2).sqrt()
        z[1] = z[1] / beta0
        q[1] = q[1] / beta0
    plane = -(-n * p // 4) * 4                                   # per-shift planes start on 16-byte boundaries
    w = [torch.zeros((S, plane), dtype=dtype, device=dev), torch.zeros((S, plane), dtype=dtype, device=dev)]
    sol = torch.zeros((S, plane), dtype=dtype, device=dev)
    scal = torch.zeros((S, 12, p), dtype=dtype, device=dev)
    scal[0, 1] = beta0
    scal[0, 11] = beta0
    scal[:, 6] = beta0                                           # scale_prev (minres.py:255)
    scal[:, 2] = 1
    scal[:, 4] = 1
    flags = torch.zeros(2, dtype=torch.int32, device=dev)
    part = torch.empty((2 * S, nb, p), dtype=dtype, device=dev)
    fold = torch.empty((lib.tsgu_cg_fold_rows(), p), dtype=dtype, device=dev)
    stream = lambda: torch.cuda.current_stream(dev).cuda_stream  # noqa: E731

    def scalar(phase, partial, rows, set_stride=0):
        _be.check(lib.tsgu_minres_scalar_ms(vt, phase, partial.data_ptr(), rows, set_stride, fold.data_ptr(), scal.data_ptr(),
                                            flags.data_ptr(), float(eps), float(settings.minres_tolerance), shifts.data_ptr(),
                                            S, val, p, dev.index, stream()), "tsgu_minres_scalar_ms")

    def vector(which, a0, a1, a2, a3=None, a4=None, qc=None, with_norms=False):
        _be.check(lib.tsgu_minres_vector_ms(vt, which, n, p, a0.data_ptr(), a1.data_ptr(), a2.data_ptr(),
                                            None if a3 is None else a3.data_ptr(), None if a4 is None else a4.data_ptr(),
                                            None if qc is None else qc.data_ptr(), scal.data_ptr(), flags.data_ptr(),
                                            part.data_ptr(), nb * p, int(with_norms), S, plane, val, dev.index, stream()),
                  "tsgu_minres_vector_ms")

    def iteration(check: bool):
        qp = z[1] if q is None else q[1]
        if fused_dot:
            prod, pqq = op.matmul_with_dot(qp)                  # A q with <q, A q> partials (minres.py:261-262)
        else:
            prod = checked(op(qp), dtype).contiguous()
            pqq = _be.coldot(prod, qp).unsqueeze(0).contiguous()
        scalar(0, pqq, pqq.shape[0])
        vector(0, z[0], z[1], prod)                             # z_c over z two steps back, |z_c|^2 partials (:263-268)
        if q is None:
            scalar(1, part[0], nb)
            vector(1, z[0], z[1], w[0], w[1], sol, with_norms=check)
        else:
            q[0] = checked(precond(z[0]), dtype).contiguous()   # q_c = M z_c, beta_c = sqrt <z_c, q_c> (:267-268)
            scalar(1, _be.coldot(z[0], q[0]).unsqueeze(0).contiguous(), 1)
            vector(1, z[0], q[1], w[0], w[1], sol, qc=q[0], with_norms=check)
            q.reverse()
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
    try_graph = fused_dot and precond is None and _graph.enabled()  # user callables are opaque: never captured
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
                  "tolerance": float(settings.minres_tolerance), "shifts": S}
    return sol[:, : n * p].reshape(S, n, p)


class _Info(threading.local):
    last = None


_INFO = _Info()


def last_solve_info():
    """Diagnostics of this thread's most recent fused ``minres`` call (iterations, stopping test outcome)."""
    return _INFO.last
