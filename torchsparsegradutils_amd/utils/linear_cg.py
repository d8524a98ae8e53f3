"""``linear_cg`` — drop-in for reference ``torchsparsegradutils/utils/linear_cg.py`` (batched
multi-RHS conjugate gradients, port of ``linear_operator``'s CG).

Same signature, settings tuple, quirks and messages: right-hand sides are normalised per column
(:257-263), convergence is the *mean* normalised residual over columns tested from iteration
``min(10, max_iter-1)`` on (:376-382), converged columns are frozen (:74), divisions are guarded
by ``eps`` (:39-43, :67-71), and a ``UserWarning`` is raised when the cap is hit (:411-421).

What changes is where the work happens.  The reference issues one sparse addmm plus ≈15 small
ATen ops and one device→host sync per iteration; here one iteration of the un-preconditioned
loop is five launches with all per-column scalars resident on the GPU:

    K1 SpMM (+ fused pᵀAp block partials) → cg_alpha → cg_update1 (r, x, rᵀr partials)
      → cg_beta (β, ‖r‖, has_converged, stop flag) → cg_update2 (p = r + βp)

and the host only polls the 4-byte stop flag every ``_POLL`` iterations (kernels turn into
no-ops once it is set, so running ahead is harmless).
"""

from __future__ import annotations

import os
import threading
import warnings
from typing import Callable, NamedTuple, Optional, Union

import torch

from .. import _backend as _be
from . import _graph
from ._operator import SparseOperator, as_operator, checked

ENABLE_FUSED_PRECOND = True  # False: preconditioned solves run the recurrences as tensor ops (tests compare the two)
_POLL = 8  # iterations enqueued between two reads of the device stop flag


class LinearCGSettings(NamedTuple):
    """Mirrors reference ``utils/linear_cg.py:10-20`` (note the default ``cg_tolerance`` of 1)."""

    max_cg_iterations: int = 1000
    max_lanczos_quadrature_iterations: int = 20
    cg_tolerance: float = 1
    terminate_cg_by_size: bool = False
    verbose_linalg: bool = False


def _default_preconditioner(x):
    return x.clone()


def _colnorm(v: torch.Tensor) -> torch.Tensor:
    """‖v‖₂ per column as a (1, p) tensor (HIP two-stage reduction)."""
    return _be.coldot(v, v).sqrt_().unsqueeze(0)


def linear_cg(
    matmul_closure: Union[torch.Tensor, Callable[[torch.Tensor], torch.Tensor]],
    rhs: torch.Tensor,
    n_tridiag: int = 0,
    tolerance: Optional[float] = None,
    eps: float = 1e-10,
    stop_updating_after: float = 1e-10,
    max_iter: Optional[int] = None,
    max_tridiag_iter: Optional[int] = None,
    initial_guess: Optional[torch.Tensor] = None,
    preconditioner: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
    settings: LinearCGSettings = LinearCGSettings(),
) -> torch.Tensor:
    r"""Solve SPD systems :math:`A x = b` for one or many right-hand sides with conjugate gradients.

    ``matmul_closure`` is a tensor (dense, or sparse COO/CSR → HIP SpMM) or a callable ``v -> A v``;
    ``rhs`` is ``(n,)`` or ``(n, k)`` (or batched ``(*batch, n, k)``) on the GPU.  Arguments and defaults mirror the
    reference.  ``n_tridiag > 0`` also returns the Lanczos tridiagonal matrices of the first ``n_tridiag`` columns,
    ``(n_tridiag, *batch, r, r)`` (reference :303-310, :385-406): those solves run the recurrences as tensor ops around the HIP
    SpMM and dot kernels (the coefficients of every iteration are needed on the way), not the fused step kernels.
    """
    if rhs.ndimension() > 2:
        return _batched_rhs(matmul_closure, rhs, n_tridiag, tolerance, eps, stop_updating_after, max_iter, max_tridiag_iter,
                            initial_guess, preconditioner, settings)
    is_vector = rhs.ndimension() == 1
    if is_vector:
        rhs = rhs.unsqueeze(-1)

    if max_iter is None:
        max_iter = settings.max_cg_iterations
    if max_tridiag_iter is None:
        max_tridiag_iter = settings.max_lanczos_quadrature_iterations
    if initial_guess is None:
        x0 = None
    else:
        is_vector = initial_guess.ndimension() == 1
        x0 = initial_guess.unsqueeze(-1) if is_vector else initial_guess
    if tolerance is None:
        tolerance = settings.cg_tolerance
    if max_tridiag_iter > max_iter:
        raise RuntimeError("Getting a tridiagonalization larger than the number of CG iterations run is not possible!")
    op = as_operator(matmul_closure)

    num_rows, p = rhs.shape
    n_iter = min(max_iter, num_rows) if settings.terminate_cg_by_size else max_iter
    n_tridiag = int(n_tridiag)
    if n_tridiag < 0 or n_tridiag > p:
        raise RuntimeError(f"n_tridiag must be between 0 and the number of right-hand sides ({p}), got {n_tridiag}")
    n_tridiag_iter = min(max_tridiag_iter, num_rows)
    dtype, dev = rhs.dtype, rhs.device
    if dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"linear_cg: unsupported dtype {dtype}")

    # normalise the right-hand sides (reference :257-263)
    rhs_norm = _colnorm(rhs)
    rhs_is_zero = rhs_norm.lt(eps)
    rhs_norm = rhs_norm.masked_fill_(rhs_is_zero, 1)
    rhs = rhs.div(rhs_norm)

    if x0 is None:
        result = torch.zeros_like(rhs)
        residual = rhs - checked(op(result), rhs.dtype)  # reference :266 (kept: it is also the NaN probe of :278)
    else:
        result = x0.div(rhs_norm).expand_as(rhs).contiguous()
        residual = rhs - checked(op(result), rhs.dtype)
    residual = residual.contiguous()

    if settings.verbose_linalg:
        print(f"Running CG on a {rhs.shape} RHS for {n_iter} iterations (tol={tolerance}). Output: {result.shape}.")

    if not torch.equal(residual, residual):
        raise RuntimeError("NaNs encountered when trying to perform matrix-vector multiplication")

    residual_norm = _colnorm(residual)
    has_converged = torch.lt(residual_norm, stop_updating_after)
    if bool(has_converged.all()) and not n_tridiag:          # (reference :286)
        n_iter = 0

    tolerance_reached = False
    k_done = 0
    t_mat = None
    if n_tridiag:
        t_mat = torch.zeros(n_tridiag_iter, n_tridiag_iter, n_tridiag, dtype=dtype, device=dev)
    fused_tridiag = None
    if (n_iter > 0 and n_tridiag and preconditioner is None and TWO_LAUNCH and p <= 256 and isinstance(op, SparseOperator) and op.dtype == dtype
            and rhs.is_cuda):
        # the coefficients of the first n_tridiag_iter iterations are recorded by the fused kernels; the matrices are built afterwards
        lib = _be.load_library()
        fused_tridiag = _two_launch_loop(lib, op, rhs_is_zero, result, residual, has_converged, n_iter, max_iter, tolerance, eps,
                                         stop_updating_after, lambda: torch.cuda.current_stream(dev).cuda_stream,
                                         n_hist=min(n_tridiag_iter, n_iter), min_iter_floor=min(n_tridiag_iter, max_iter - 1))
    if fused_tridiag is not None:
        result, residual_norm, k_done, tolerance_reached, hist = fused_tridiag
        # (reference: the iteration that meets the stop rule leaves before its own row is written)
        rows = min(n_tridiag_iter, k_done - 1 if tolerance_reached else k_done)
        last_tridiag_iter = _tridiag_from_history(hist, n_tridiag, rows, t_mat)
    elif n_iter > 0:
        # (CPU operands: the recurrences as tensor ops — the fused step kernels are the MI355X path)
        if n_tridiag or (preconditioner is not None and not ENABLE_FUSED_PRECOND) or not rhs.is_cuda:
            result, residual_norm, k_done, tolerance_reached, last_tridiag_iter = _pcg_loop(
                op, preconditioner, rhs_is_zero, result, residual, has_converged, n_iter, max_iter, tolerance, eps,
                stop_updating_after, n_tridiag, n_tridiag_iter, t_mat,
            )
        else:
            last_tridiag_iter = 0
            result, residual_norm, k_done, tolerance_reached = _fused_loop(
                op, rhs_is_zero, result, residual, has_converged, n_iter, max_iter, tolerance, eps,
                stop_updating_after, preconditioner,
            )

    result = result.mul(rhs_norm)

    if not tolerance_reached and n_iter > 0:
        warnings.warn(
            "CG terminated in {} iterations with average residual norm {}"
            " which is larger than the tolerance of {} specified by"
            " linear_operator.settings.cg_tolerance."
            " If performance is affected, consider raising the maximum number of CG iterations by running code in"
            " a linear_operator.settings.max_cg_iterations(value) context.".format(
                k_done, residual_norm.mean(), tolerance
            ),
            UserWarning,
        )

    _INFO.last = {"solver": "linear_cg", "iterations": int(k_done), "tolerance_reached": bool(tolerance_reached) or n_iter == 0,
                  "residual_norm": residual_norm.detach().reshape(-1).clone(), "tolerance": float(tolerance)}
    if settings.verbose_linalg:
        print(f"CG finished after {k_done} iterations; mean normalised residual {float(residual_norm.mean()):.3e} "
              f"(tolerance {tolerance}, reached={bool(tolerance_reached) or n_iter == 0}).")

    if is_vector:
        result = result.squeeze(-1)
    if n_tridiag:
        last = last_tridiag_iter if n_iter > 0 else 0
        t_mat = t_mat[: last + 1, : last + 1]
        return result, t_mat.permute(-1, 0, 1).contiguous()       # (reference :426-427)
    return result


class _Info(threading.local):
    last = None


_INFO = _Info()


def last_solve_info():
    """Diagnostics of this thread's most recent ``linear_cg`` call: iterations executed, whether the tolerance
    was reached and the final rhs-normalised residual norm per column (the device already holds them; the
    reference only offers ``verbose_linalg`` printing, utils/linear_cg.py:273-275)."""
    return _INFO.last


def _batched_rhs(matmul_closure, rhs, n_tridiag, tolerance, eps, stop_updating_after, max_iter, max_tridiag_iter,
                 initial_guess, preconditioner, settings):
    """Right-hand sides with batch dimensions ``(*batch, n, k)`` (reference utils/linear_cg.py:257-263, :378: norms
    and convergence are per (batch, column), the stop rule is their mean): the batch is folded into the columns,
    ``(n, batch·k)``, and solved by the 2-D path.  A 2-D sparse operator applies to every column alike; any other
    closure sees its own ``(*batch, n, k)`` layout through a reshaping wrapper."""
    batch_shape = tuple(rhs.shape[:-2])
    n, k = rhs.shape[-2:]
    nb = 1
    for d in batch_shape:
        nb *= d

    def fold(t):      # (*batch, n, k) -> (n, batch*k)
        return t.reshape(nb, n, k).permute(1, 0, 2).reshape(n, nb * k)

    def unfold(t):    # (n, batch*k) -> (*batch, n, k)
        return t.reshape(n, nb, k).permute(1, 0, 2).reshape(batch_shape + (n, k))

    def wrap(fn):
        return lambda v: fold(fn(unfold(v)))

    if torch.is_tensor(matmul_closure) and matmul_closure.layout in (torch.sparse_csr, torch.sparse_coo):
        op = matmul_closure
    else:
        op = wrap(as_operator(matmul_closure))
    x0 = None if initial_guess is None else fold(initial_guess.expand(rhs.shape))
    pre = None if preconditioner is None else wrap(preconditioner)
    if not n_tridiag:
        out = linear_cg(op, fold(rhs).contiguous(), 0, tolerance, eps, stop_updating_after, max_iter, max_tridiag_iter, x0, pre, settings)
        return unfold(out)
    # Lanczos coefficients of the first n_tridiag columns of EVERY batch item (reference :303-310: its tridiagonal bookkeeping —
    # the `t_mat[k-1, k].max() < 1e-6` rule that ends it early, and with it the size of T — looks at batch x n_tridiag columns
    # only).  The folded columns are reordered so that those columns lead, the 2-D path tridiagonalises exactly them, and the
    # solution is put back in the caller's column order.
    if n_tridiag > k:
        raise RuntimeError(f"n_tridiag must be between 0 and the number of right-hand sides ({k}), got {n_tridiag}")
    dev = rhs.device
    col = torch.arange(nb * k, device=dev).reshape(nb, k)
    order = torch.cat((col[:, :n_tridiag].reshape(-1), col[:, n_tridiag:].reshape(-1)))      # leading: (item, first n_tridiag columns)
    back = torch.empty_like(order)
    back[order] = torch.arange(nb * k, device=dev)

    def lead(fn):     # an operator on folded columns in the reordered layout
        return lambda v: fn(v[:, back])[:, order]

    if op is not matmul_closure:
        op = lead(op)
    x0r = None if x0 is None else x0[:, order].contiguous()
    prer = None if pre is None else lead(pre)
    out, t_lead = linear_cg(op, fold(rhs)[:, order].contiguous(), nb * n_tridiag, tolerance, eps, stop_updating_after, max_iter,
                            max_tridiag_iter, x0r, prer, settings)
    r = t_lead.shape[-1]
    t_sel = t_lead.reshape(nb, n_tridiag, r, r)                             # (batch, n_tridiag, r, r)
    return unfold(out[:, back]), t_sel.permute(1, 0, 2, 3).reshape((n_tridiag,) + batch_shape + (r, r)).contiguous()


TWO_LAUNCH = os.environ.get("TSGU_CG_TWO_LAUNCH", "1") != "0"


class _PollCache(threading.local):
    bufs = None


_POLL_CACHE = _PollCache()


def _poll_buffers(dev):
    """Two (pinned int32[2], event) pairs per thread and device for the host's polls (allocating pinned memory costs more than a
    short solve's iterations)."""
    if _POLL_CACHE.bufs is None:
        _POLL_CACHE.bufs = {}
    got = _POLL_CACHE.bufs.get(dev)
    if got is None:
        with torch.cuda.device(dev):
            got = _POLL_CACHE.bufs[dev] = [(torch.empty(2, dtype=torch.int32, pin_memory=True), torch.cuda.Event()) for _ in range(2)]
    return got


def _two_launch_loop(lib, op, rhs_is_zero, x, r, has_converged, n_iter, max_iter, tolerance, eps, stop_after, stream, n_hist=0,
                     min_iter_floor=0):
    """The iterations as K1 (+ p'Ap partials) -> tsgu_cg2_residual -> tsgu_cg2_direction (include/tsgu_hip.h): the state an
    iteration reads sits in the half of its parity, what it produces goes to the other half, so no single-workgroup kernel is
    left between the streaming ones.  Returns None when K1 leaves more partial rows than the kernels sum per workgroup (the
    caller then runs the four-step form).  Same recurrences, same stop rule (reference :319-382).  `n_hist` > 0: alpha and beta of
    the first n_hist iterations are kept ([n_hist][2][p], returned as a fifth value) — what the Lanczos tridiagonal matrices are
    built from; `min_iter_floor`: no stop before that many iterations (reference :379: not while the matrices are being filled)."""
    n, p = r.shape
    dev, dtype = r.device, r.dtype
    vt = _be.vtype_of(r)
    nb = lib.tsgu_cg2_num_blocks(vt, n, p)
    if nb <= 0:
        return None
    pvec = r.clone()  # curr_conjugate_vec (reference :293)
    Ap, pap = op.matmul_with_dot(pvec)            # the first product also tells how many partial rows K1 leaves
    if pap.shape[0] > 1024 or not pap.is_contiguous():
        return None
    scal = torch.zeros(5 * p, dtype=dtype, device=dev)
    scal[:p] = _be.coldot(r, r)  # residual_inner_prod (reference :294)
    flags = torch.zeros(4 + 3 * p, dtype=torch.int32, device=dev)
    flags[4 : 4 + p] = has_converged.reshape(-1).to(torch.int32)
    flags[4 + 2 * p :] = rhs_is_zero.reshape(-1).to(torch.int32)
    rr_partial = torch.empty((nb, p), dtype=dtype, device=dev)
    min_iter_index = max(min(10, max_iter - 1), min_iter_floor)
    hist = torch.zeros((n_hist, 2, p), dtype=dtype, device=dev) if n_hist > 0 else None
    hist_addr = hist.data_ptr() if hist is not None else None
    state = {"parity": 0, "first": (Ap, pap)}

    flags_addr = flags.data_ptr()

    def iteration():
        par = state["parity"]
        if state["first"] is not None:
            Ap, pap = state["first"]
            state["first"] = None
        else:
            # K1 + p'Ap partials (reference :322, :64-65); it does nothing once this half's done word is set
            Ap, pap = op.matmul_with_dot(pvec, skip=flags_addr + 4 * par)
        s = stream()
        _be.check(lib.tsgu_cg2_residual(vt, n, p, r.data_ptr(), Ap.data_ptr(), pap.data_ptr(), pap.shape[0], scal.data_ptr(),
                                        flags_addr, par, eps, rr_partial.data_ptr(), dev.index, s), "tsgu_cg2_residual")
        _be.check(lib.tsgu_cg2_direction(vt, n, p, r.data_ptr(), pvec.data_ptr(), x.data_ptr(), rr_partial.data_ptr(), nb,
                                         scal.data_ptr(), flags_addr, par, eps, stop_after, float(tolerance), min_iter_index,
                                         hist_addr, n_hist, dev.index, s), "tsgu_cg2_direction")
        state["parity"] = par ^ 1

    # The host polls one chunk BEHIND the device: after queueing chunk j it copies the done words to pinned memory (asynchronously,
    # behind chunk j in the stream), queues chunk j + 1 and only then waits for the copy of chunk j — the GPU always has the next
    # chunk in its queue (a blocking read per chunk left it idle for a launch + a round trip: ~5 us per iteration at C4).
    # Iterations queued past the end are no-ops on the device (K1 included: `skip`).
    done = False
    k = 0
    graph = None
    try_graph = _graph.enabled()
    polls = _poll_buffers(dev)
    pending = None
    which = 0
    with torch.cuda.device(dev):
        while k < n_iter and not done:
            if try_graph and graph is None and k > min_iter_index and n_iter - k >= _graph.MIN_ITERS:
                # (an even number of iterations per replay: the parities baked into the captured launches stay right)
                # (a capture that fails part-way has run nothing on the device: the host's view of the parity must not move either)
                saved = dict(state)
                graph = _graph.capture(iteration, _POLL)
                try_graph = graph is not None
                if graph is None:
                    state.update(saved)
            if graph is not None and k + _POLL <= n_iter:
                _graph.replay(graph)
                k += _POLL
            else:
                upto = min(n_iter, max(k + _POLL, min_iter_index + 1) if k <= min_iter_index else k + _POLL)
                for _ in range(k, upto):
                    iteration()
                k = upto
            buf, ev = polls[which]
            buf.copy_(flags[:2], non_blocking=True)
            ev.record()
            if pending is not None:
                pending[1].synchronize()
                done = bool(pending[0][0] != 0 or pending[0][1] != 0)
            pending = polls[which]
            which ^= 1
        head = flags[:2].tolist()   # (waits for everything queued)
        done = head[0] != 0 or head[1] != 0
    k_done = int(flags[2].item())
    rnorm = scal[4 * p : 5 * p].unsqueeze(0)
    if n_hist > 0:
        return x, rnorm, k_done, done, hist
    return x, rnorm, k_done, done


def _tridiag_from_history(hist: torch.Tensor, n_tridiag: int, rows: int, t_mat: torch.Tensor) -> int:
    """Lanczos tridiagonal matrices from the CG coefficients of the first `rows` iterations (reference :385-406, the same
    arithmetic on the recorded alpha / beta): fills `t_mat` [n_tridiag_iter][n_tridiag_iter][n_tridiag] and returns the last
    row written.  The reference stops filling once an off-diagonal row falls below 1e-6 in every column — that row included."""
    if rows <= 0:
        return 0
    alpha = hist[:rows, 0, :n_tridiag]
    beta = hist[:rows, 1, :n_tridiag]
    ar = torch.where(alpha == 0, torch.ones_like(alpha), alpha).reciprocal()
    diag = ar.clone()
    diag[1:] += beta[:-1] * ar[:-1]
    off = beta[:-1].sqrt() * ar[:-1]                       # off[k-1] = t[k, k-1], k = 1 .. rows-1
    last = rows - 1
    if rows > 1:
        small = (off.amax(dim=1) < 1e-6).nonzero()
        if small.numel():
            last = int(small[0]) + 1                        # (the row whose off-diagonal entry is small is still written)
    k = torch.arange(last + 1, device=hist.device)
    t_mat[k, k] = diag[: last + 1]
    if last >= 1:
        t_mat[k[1:], k[:-1]] = off[:last]
        t_mat[k[:-1], k[1:]] = off[:last]
    return last


def _fused_loop(op, rhs_is_zero, x, r, has_converged, n_iter, max_iter, tolerance, eps, stop_after, preconditioner=None):
    """CG iterations on the fused gfx950 kernels (reference :319-382, :50-95).  A preconditioner is called between the
    residual update and the beta step (z = M r); the recurrences then run on <r, z>, the stop test on |r|."""
    lib = _be.load_library()
    n, p = r.shape
    dev, dtype = r.device, r.dtype
    vt = _be.vtype_of(r)
    nb_upd = lib.tsgu_cg_num_blocks(vt, n, p)
    if nb_upd < 0:
        raise RuntimeError("linear_cg: more than 1024 simultaneous right-hand sides are not supported")

    fused_dot = isinstance(op, SparseOperator) and op.dtype == dtype
    stream = lambda: torch.cuda.current_stream(dev).cuda_stream  # noqa: E731
    if preconditioner is None and fused_dot and p <= 256 and TWO_LAUNCH:
        got = _two_launch_loop(lib, op, rhs_is_zero, x, r, has_converged, n_iter, max_iter, tolerance, eps, stop_after, stream)
        if got is not None:
            return got

    # device state: scal = [rr | alpha | beta | rnorm], flags = [done, iters, has_converged[p], rhs_is_zero[p]]
    scal = torch.zeros(4 * p, dtype=dtype, device=dev)
    if preconditioner is None:
        scal[:p] = _be.coldot(r, r)  # residual_inner_prod (reference :294)
        pvec = r.clone()  # curr_conjugate_vec (reference :293)
    else:
        z = checked(preconditioner(r), dtype).contiguous()  # (reference :291-294)
        scal[:p] = _be.coldot(z, r)
        pvec = z.clone() if z.data_ptr() == r.data_ptr() else z
    flags = torch.zeros(2 + 2 * p, dtype=torch.int32, device=dev)
    flags[2 : 2 + p] = has_converged.reshape(-1).to(torch.int32)
    flags[2 + p :] = rhs_is_zero.reshape(-1).to(torch.int32)
    rr_partial = torch.empty((nb_upd, p), dtype=dtype, device=dev)
    fold = torch.empty((lib.tsgu_cg_fold_rows(), p), dtype=dtype, device=dev)
    min_iter_index = min(10, max_iter - 1)

    def iteration():
        if fused_dot:
            Ap, pap = op.matmul_with_dot(pvec)  # K1 + pᵀAp partials (reference :322, :64-65)
            n_partial = pap.shape[0]
        else:
            Ap = checked(op(pvec), dtype).contiguous()
            pap = _be.coldot(pvec, Ap).unsqueeze(0)
            n_partial = 1
        s = stream()
        if n_partial <= 1024 and p <= 256 and pap.is_contiguous():
            # few partial rows (K1 on the plane sweep, or an operator with its own dot): alpha is summed by every workgroup of the
            # update itself — one launch and one single-workgroup kernel less per iteration
            _be.check(lib.tsgu_cg_update1_alpha(vt, n, p, r.data_ptr(), Ap.data_ptr(), x.data_ptr(), pvec.data_ptr(), pap.data_ptr(),
                                                n_partial, scal.data_ptr(), flags.data_ptr(), eps, rr_partial.data_ptr(), dev.index, s),
                      "tsgu_cg_update1_alpha")
        else:
            _be.check(lib.tsgu_cg_alpha(vt, pap.data_ptr(), n_partial, fold.data_ptr(), scal.data_ptr(), flags.data_ptr(), eps, p,
                                        dev.index, s), "tsgu_cg_alpha")
            _be.check(lib.tsgu_cg_update1(vt, n, p, r.data_ptr(), Ap.data_ptr(), x.data_ptr(), pvec.data_ptr(),
                                          scal.data_ptr(), flags.data_ptr(), rr_partial.data_ptr(), dev.index, s),
                      "tsgu_cg_update1")
        # iteration index -1: the counter is flags[1] on the device, every iteration is the same launch
        if preconditioner is None:
            _be.check(lib.tsgu_cg_beta(vt, rr_partial.data_ptr(), nb_upd, scal.data_ptr(), flags.data_ptr(), eps,
                                       stop_after, float(tolerance), -1, min_iter_index, p, dev.index, s),
                      "tsgu_cg_beta")
            z = r
        else:
            z = checked(preconditioner(r), dtype).contiguous()  # z = M r (reference :80)
            rz = _be.coldot(z, r)
            _be.check(lib.tsgu_cg_beta_precond(vt, rr_partial.data_ptr(), nb_upd, rz.data_ptr(), 1, scal.data_ptr(), flags.data_ptr(),
                                               eps, stop_after, float(tolerance), -1, min_iter_index, p, dev.index, s),
                      "tsgu_cg_beta_precond")
        _be.check(lib.tsgu_cg_update2(vt, n, p, z.data_ptr(), pvec.data_ptr(), scal.data_ptr(),
                                      flags.data_ptr(), dev.index, s), "tsgu_cg_update2")

    done = False
    k = 0
    graph = None
    try_graph = fused_dot and preconditioner is None and _graph.enabled()  # user callables are opaque (may synchronise): never captured
    with torch.cuda.device(dev):
        while k < n_iter and not done:
            if try_graph and graph is None and k > min_iter_index and n_iter - k >= _graph.MIN_ITERS:
                # long solves: capture _POLL iterations once as a hipGraph and replay it (one host call per
                # chunk instead of 6 launches per iteration; finished iterations are device-side no-ops)
                graph = _graph.capture(iteration, _POLL)
                try_graph = graph is not None
            if graph is not None and k + _POLL <= n_iter:
                _graph.replay(graph)
                k += _POLL
            else:
                upto = min(n_iter, max(k + _POLL, min_iter_index + 1) if k <= min_iter_index else k + _POLL)
                for _ in range(k, upto):
                    iteration()
                k = upto
            head = flags[:2].tolist()  # the only device→host read: [done, iterations executed]
            done = head[0] != 0
    k_done = int(flags[1].item())
    rnorm = scal[3 * p : 4 * p].unsqueeze(0)
    return x, rnorm, k_done, done


def _pcg_loop(op, preconditioner, rhs_is_zero, x, r, has_converged, n_iter, max_iter, tolerance, eps, stop_after,
              n_tridiag=0, n_tridiag_iter=0, t_mat=None):
    """Iterations with the recurrences as device tensor ops (reference :319-406): preconditioned solves — the user's
    preconditioner is an opaque callable — and solves that also return Lanczos tridiagonal matrices, which need alpha and beta
    of every iteration.  SpMM and dots stay on the HIP kernels."""
    dtype = r.dtype
    if preconditioner is None:
        preconditioner = _default_preconditioner
    z = preconditioner(r)
    pvec = z.clone() if z is r else z
    rz = _be.coldot(z.contiguous(), r).unsqueeze(0)
    rnorm = _colnorm(r)
    done = False
    k_done = 0
    eps_t = torch.tensor(eps, dtype=dtype, device=r.device)
    update_tridiag, last_tridiag_iter = True, 0
    prev_alpha_reciprocal = prev_beta = None
    for k in range(n_iter):
        Ap = checked(op(pvec), dtype).contiguous()
        pap = _be.coldot(pvec.contiguous(), Ap).unsqueeze(0)
        zero = pap < eps_t
        alpha = torch.where(zero, torch.zeros_like(pap), rz / torch.where(zero, torch.ones_like(pap), pap))
        alpha = alpha.masked_fill(has_converged, 0)
        r = torch.addcmul(r, alpha, Ap, value=-1)
        z = preconditioner(r)
        x = torch.addcmul(x, alpha, pvec)
        rz_old = rz
        rz = _be.coldot(z.contiguous(), r.contiguous()).unsqueeze(0)
        zero = rz_old < eps_t
        beta = torch.where(zero, torch.zeros_like(rz), rz / torch.where(zero, torch.ones_like(rz), rz_old))
        pvec = pvec * beta + z
        rnorm = _colnorm(r.contiguous()).masked_fill_(rhs_is_zero, 0)
        has_converged = rnorm < stop_after
        k_done = k + 1
        if (k >= min(10, max_iter - 1) and bool(rnorm.mean() < tolerance)
                and not (n_tridiag and k < min(n_tridiag_iter, max_iter - 1))):
            done = True
            break
        # Lanczos tridiagonal matrices from the CG coefficients (reference :385-406)
        if n_tridiag and k < n_tridiag_iter and update_tridiag:
            alpha_tridiag = alpha.squeeze(0)[:n_tridiag]
            beta_tridiag = beta.squeeze(0)[:n_tridiag]
            alpha_is_zero = alpha_tridiag == 0
            alpha_reciprocal = torch.where(alpha_is_zero, torch.ones_like(alpha_tridiag), alpha_tridiag).reciprocal()
            if k == 0:
                t_mat[k, k].copy_(alpha_reciprocal)
            else:
                t_mat[k, k] = alpha_reciprocal + prev_beta * prev_alpha_reciprocal
                t_mat[k, k - 1] = prev_beta.sqrt() * prev_alpha_reciprocal
                t_mat[k - 1, k].copy_(t_mat[k, k - 1])
                if float(t_mat[k - 1, k].max()) < 1e-6:
                    update_tridiag = False
            last_tridiag_iter = k
            prev_alpha_reciprocal = alpha_reciprocal
            prev_beta = beta_tridiag.clone()
    return x, rnorm, k_done, done, last_tridiag_iter
