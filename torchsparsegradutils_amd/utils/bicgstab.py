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

import threading

import torch

from .. import _backend as _be
from . import _graph
from ._operator import SparseOperator, as_operator, checked

ENABLE_FUSED = True  # False: the reference's op chain around the K1 matvec, column by column (tests compare the two)
_POLL = 4  # iterations enqueued between two reads of the device "all columns finished" word
_GRAPH_AFTER = 16  # iterations run eagerly before a chunk is recorded as a hipGraph

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
    if settings.precon is not None and not (torch.is_tensor(settings.precon) or callable(settings.precon)):
        raise RuntimeError("settings.precon must be a tensor, or a callable object!")
    if ENABLE_FUSED and rhs.is_cuda and rhs.dtype in (torch.float32, torch.float64) and rhs.dim() in (1, 2) \
            and rhs.shape[-1 if rhs.dim() == 2 else 0] > 0 and (rhs.dim() == 1 or rhs.shape[1] <= 1024):
        return _bicgstab_fused(matmul_closure, rhs, initial_guess, settings)
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
    else:
        precon = settings.precon

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


def _bicgstab_fused(matmul_closure, rhs, initial_guess, settings: BICGSTABSettings) -> torch.Tensor:
    """All columns in lock-step on the K6 kernels (csrc/bicgstab.hip); same per-column arithmetic and stopping
    rules as the reference's column loop (utils/bicgstab.py:126-247).  A preconditioner (settings.precon, :191-194,
    :216-219) is applied between the kernels: a tensor to all columns at once, a callable column by column on
    contiguous vectors, as the reference calls it."""
    lib = _be.load_library()
    is_vector = rhs.dim() == 1
    B = (rhs.unsqueeze(-1) if is_vector else rhs).contiguous()
    n, p = B.shape
    dev, dtype = B.device, B.dtype
    vt = _be.vtype_of(B)
    op = as_operator(matmul_closure)
    fused_dot = isinstance(op, SparseOperator) and op.dtype == dtype
    matvec_max = 2 * n if settings.matvec_max is None else int(settings.matvec_max)
    matvec_max = min(matvec_max, 2**31 - 1)
    if settings.precon is None:
        precon = None
    elif torch.is_tensor(settings.precon):
        m_op = as_operator(settings.precon)
        precon = lambda V: checked(m_op(V), dtype).contiguous()  # noqa: E731
    else:
        m_fn = settings.precon
        precon = lambda V: checked(torch.stack([m_fn(V[:, j].contiguous()) for j in range(p)], dim=1), dtype)  # noqa: E731

    if initial_guess is None:
        x = torch.zeros_like(B)
        r0 = (B - checked(op(x), dtype)).contiguous()  # one matvec on A·0, as the reference (bicgstab.py:159-161)
        nmv0 = 1
    else:
        x = (initial_guess.unsqueeze(-1) if initial_guess.dim() == 1 else initial_guess).clone().contiguous()
        r0 = B.clone()  # the reference does not subtract A·x0 (bicgstab.py:158)
        nmv0 = 0

    nb = lib.tsgu_cg_num_blocks(vt, n, p)
    if nb < 0:
        raise RuntimeError("bicgstab: more than 1024 simultaneous right-hand sides are not supported")
    scal = torch.zeros(8 * p, dtype=dtype, device=dev)
    flags = torch.zeros(2 + 3 * p, dtype=torch.int32, device=dev)
    part = torch.empty((3, nb, p), dtype=dtype, device=dev)
    fold = torch.empty((lib.tsgu_cg_fold_rows(), p), dtype=dtype, device=dev)
    stream = lambda: torch.cuda.current_stream(dev).cuda_stream  # noqa: E731

    def scalar(phase, partial, rows, set_stride=0):
        _be.check(lib.tsgu_bicg_scalar(vt, phase, None if partial is None else partial.data_ptr(), rows, set_stride,
                                       fold.data_ptr(), scal.data_ptr(), flags.data_ptr(), float(settings.abstol),
                                       float(settings.reltol), matvec_max, nmv0, p, dev.index, stream()), "tsgu_bicg_scalar")

    def vector(which, a0, a1, a2, a3=None, a4=None, partial=None, set_stride=0):
        _be.check(lib.tsgu_bicg_vector(vt, which, n, p, a0.data_ptr(), a1.data_ptr(), a2.data_ptr(),
                                       None if a3 is None else a3.data_ptr(), None if a4 is None else a4.data_ptr(),
                                       scal.data_ptr(), flags.data_ptr(), None if partial is None else partial.data_ptr(),
                                       set_stride, dev.index, stream()), "tsgu_bicg_vector")

    with torch.cuda.device(dev):
        rr0 = _be.coldot(r0, r0).unsqueeze(0).contiguous()
        scalar(0, rr0, 1)
        r = r0.clone()
        pv = torch.zeros_like(B)
        v = torch.zeros_like(B)
        s = torch.zeros_like(B)

        def iteration():
            scalar(1, None, 0)                      # beta, rho (bicgstab.py:183-184)
            vector(0, pv, r, v)                     # p update (:187-189)
            q = pv if precon is None else precon(pv)  # (:191-194)
            if fused_dot:
                _, pr0v = op.matmul_with_dot(q, r0, out=v)  # v = A q with <r0, v> partials (:196-199)
                scalar(2, pr0v, pr0v.shape[0])
            else:
                v.copy_(checked(op(q), dtype))
                scalar(2, _be.coldot(r0, v).unsqueeze(0).contiguous(), 1)
            vector(1, s, r, v, partial=part[0])     # s = r - alpha v, |s|^2 (:200-203)
            scalar(3, part[0], nb)                  # early exit / matvec budget (:207-214)
            z = s if precon is None else precon(s)  # (:216-219)
            t = checked(op(z), dtype).contiguous()                  # t = A z (:221)
            vector(2, t, s, r0, partial=part, set_stride=nb * p)
            scalar(4, part, nb, nb * p)             # omega, rho_next (:223-224)
            if precon is None:
                vector(3, x, r, s, t, pv, partial=part[0])  # r, x updates, |r|^2 (:227-235)
            else:
                _be.check(lib.tsgu_bicg_update_x_precond(vt, n, p, x.data_ptr(), r.data_ptr(), s.data_ptr(), t.data_ptr(),
                                                         q.data_ptr(), z.data_ptr(), scal.data_ptr(), flags.data_ptr(),
                                                         part[0].data_ptr(), dev.index, stream()), "tsgu_bicg_update_x_precond")
            scalar(5, part[0], nb)                  # stop tests (:239-241)

        done = bool(flags[0].item())
        k = 0
        graph = None
        # user callables (operator or preconditioner) are opaque (may synchronise): never captured
        try_graph = fused_dot and _graph.enabled() and (precon is None or torch.is_tensor(settings.precon))
        while not done:
            if try_graph and graph is None and k >= _GRAPH_AFTER and (matvec_max - nmv0) // 2 - k >= _graph.MIN_ITERS:
                # still running after _GRAPH_AFTER iterations: record one chunk as a hipGraph and replay it
                graph = _graph.capture(iteration, _POLL)
                try_graph = graph is not None
            if graph is not None:
                _graph.replay(graph)
            else:
                for _ in range(_POLL):
                    iteration()
            k += _POLL
            done = bool(flags[0].item())
    _INFO.last = {"solver": "bicgstab", "iterations_enqueued": k, "finished": True}
    return x.squeeze(-1) if is_vector else x


class _Info(threading.local):
    last = None


_INFO = _Info()


def last_solve_info():
    """Diagnostics of this thread's most recent fused ``bicgstab`` call."""
    return _INFO.last
