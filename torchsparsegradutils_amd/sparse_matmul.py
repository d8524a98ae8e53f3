"""``sparse_mm`` / ``SparseMatMul`` — drop-in for reference ``torchsparsegradutils/sparse_matmul.py``.

Same signature, validation order and messages (reference :114-127), same autograd contract
(gradient of the sparse operand has A's layout, A's index tensors and A's index dtype;
``needs_input_grad`` gating; saved tensors are released after the first backward).  The
arithmetic runs in the hand-written gfx950 kernels:

=============================  ================================  =====================
reference (ATen)               here                              C ABI
=============================  ================================  =====================
``torch.sparse.mm(A, B)``      K1 CSR SpMM            (:155)     ``tsgu_csr_spmm``
gathers + mul + sum  :186-205  K3 fused SDDMM                    ``tsgu_csr_sddmm``
``torch.sparse.mm(A.t(), G)``  K2 gather SpMM on the cached      ``tsgu_csr_spmm``
:229                           transposed pattern                (``perm`` argument)
block-diag assembly  :151-153  none: batched CSR stays batched   ``batch`` argument
=============================  ================================  =====================
"""

from __future__ import annotations

import ctypes
import os
from typing import cast

import torch

from . import _backend as _be
from . import _ops
from . import _pattern as _pt
from ._tile import TilePlanStruct as _TilePlanStruct


def sparse_mm(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    r"""Sparse–dense matrix product :math:`C = A B` with sparsity-preserving gradients.

    ``A``: sparse COO or CSR, ``(n, m)`` or ``(b, n, m)``; ``B``: dense ``(m, p)`` or ``(b, m, p)``.
    Returns dense ``(n, p)`` / ``(b, n, p)``.  ``dL/dA`` is evaluated only at the stored entries
    of ``A`` (:math:`[\partial A]_{ij} = \langle G_{i,:}, B_{j,:}\rangle`) and returned with A's
    layout; ``dL/dB = A^\top G``.  Mirrors reference ``sparse_matmul.py:8-129``.
    """
    if not isinstance(A, torch.Tensor) or not isinstance(B, torch.Tensor):
        raise ValueError("Both A and B should be instances of torch.Tensor")
    if A.dim() < 2 or B.dim() < 2:
        raise ValueError("Both A and B should be at least 2-dimensional tensors")
    if A.dim() != B.dim() or A.dim() not in (2, 3):
        raise ValueError("A and B must both be 2D or both be 3D tensors")
    if A.layout not in {torch.sparse_coo, torch.sparse_csr}:
        raise ValueError("A should be in either COO or CSR sparse format")
    if B.layout != torch.strided:
        raise ValueError("B must be a dense (strided) tensor")
    if A.dim() == 3 and A.size(0) != B.size(0):
        raise ValueError("If batched, A and B must have the same batch size")
    if A.size(-1) != B.size(-2):
        raise ValueError(f"Incompatible inner dimensions: A[..., {A.size(-1)}] vs B[..., {B.size(-2)}]")

    if _host is not None and FAST_STEP and B.is_cuda and (B.dim() == 2 or A.layout == torch.sparse_csr):
        if SPECULATE and A.layout == torch.sparse_csr:
            C = _speculative_step(A, B)
            if C is not None:
                return C
        plan = _step_plan(A, B)
        if plan is not None:
            return cast(torch.Tensor, _host.step(A, B, plan))
    return cast(torch.Tensor, SparseMatMul.apply(A, B))


# ---- steady state: the step's host path in C++ (csrc/host/step.cpp) -----------------------------------------------------------
# Once the three launch configurations of a CSR pattern are final (plane march / plane sweep, found by the Python path below during
# the first steps), forward and backward are the same three launches through the same C ABI, issued by a torch::autograd::Function
# written in C++: no interpreter on the autograd engine's thread, no ctypes marshalling.  A fast host spends 0.08 ms per step in
# the Python path, a slow one 0.25 ms — more than the 0.23 ms the kernels of a C2 step take.  TSGU_FAST_STEP=0 keeps the Python path.
FAST_STEP = os.environ.get("TSGU_FAST_STEP", "1") != "0"
try:
    from . import _tsgu_host as _host       # built by csrc/Makefile next to this file
except ImportError:                          # (the Python path below is complete by itself; the kernels are the same)
    _host = None
if os.environ.get("TSGU_LIB_PATH"):          # another build of the kernels is loaded (A/B experiments): _tsgu_host.so is linked against
    _host = None                             # csrc/libtsgu_hip.so and would launch THAT build's kernels


# A caller that rebuilds its index tensors every step misses the pattern cache's identity key; whether the fresh tensors hold a known
# pattern is one pass over them + one host read (`_pattern._core_for`).  With a live pattern of the same geometry that has a step
# plan, the forward of THAT plan is queued right behind the comparison pass, before its answer is read: the host's reaction to the
# answer (~50 us) then overlaps the forward kernel instead of an idle GPU.  Equal (the common case): the step was the right one.
# Other content: the queued result is dropped — never returned — and the normal path runs on the new pattern.
SPECULATE = os.environ.get("TSGU_SPECULATE_FRESH", "1") != "0"


def _speculative_step(A: torch.Tensor, B: torch.Tensor):
    spec = _pt.speculate_csr(A)
    if spec is None:
        return None
    C = None
    try:
        if (A.dtype == B.dtype and A.device == B.device and B.is_contiguous() and B.data_ptr() % 16 == 0 and _be.KERNEL_EVENTS is None
                and not torch.cuda.is_current_stream_capturing()):
            plans = spec.candidate.own.get("step_plans")
            sp = None if plans is None else plans.get(_step_key(B.dtype, B.size(-1)))
            if sp is not None:
                C = _host.step(A, B, sp)
    except RuntimeError:
        C = None                           # (the candidate's plan does not take these operands: the normal path decides)
    finally:
        same = spec.finish()               # (always: the cache entry of these tensors exists from here on, adopted or new)
    return cast(torch.Tensor, C) if same and C is not None else None


def _step_key(dtype, p: int):
    return (dtype, p, _ops.ENABLE_LATTICE, _ops._lt.ENABLE_MARCH, _ops.ENABLE_PACK, _ops.ENABLE_TILE)


def _step_plan(A: torch.Tensor, B: torch.Tensor):
    """The `_tsgu_host.StepPlan` of (A's pattern, B's dtype and width), or None while the pattern is young / not covered."""
    if A.dtype != B.dtype or A.device != B.device or not B.is_contiguous() or B.data_ptr() % 16 or _be.KERNEL_EVENTS is not None:
        return None
    if A.layout == torch.sparse_csr:
        own = _pt.from_csr(A).core.own
    elif A.is_coalesced():
        own = _pt.from_coo_2d(A._indices(), A.shape, coalesced=True).core.own
    else:
        return None
    plans = own.get("step_plans")
    if plans is None:
        return None
    return plans.get(_step_key(B.dtype, B.size(-1)))


# consecutive steps a pattern has to launch the very same plans before the C++ host path takes the step over
SETTLE_AFTER = 2


def _settle_step_plan(op: "_Operand", values: torch.Tensor, G: torch.Tensor, B: torch.Tensor) -> None:
    """After a step on the Python path (both gradients): when what the step launches has stopped changing, describe it to the C++ host
    path (csrc/host/step.cpp).  Nothing is DECIDED here: `_ops.spmm` / `_ops.mm_backward` note the family and the plan objects they
    launched with the pattern (`_ops.launched`); this function only waits until forward and backward have made the same choice
    SETTLE_AFTER steps in a row with no plan build in flight, and then wraps exactly those objects — the three configurations of a
    lattice stencil (the StepPlan copies the plan structs and holds every device table they point into), the two tile plans, or the
    plan-free kernels with the cached transposed pattern."""
    plan = op.plan
    if (_host is None or not FAST_STEP or not B.is_cuda or not plan.crow.is_cuda or plan.perm is not None or op.flat_batch is not None
            or not (values.dtype == G.dtype == B.dtype) or (op.layout != torch.sparse_csr and op.indices is None)):
        return
    dtype, p = G.dtype, G.size(-1)
    own = plan.core.own
    batched = plan.batch is not None
    if B.dim() != (3 if batched else 2) or not (B.is_contiguous() and G.is_contiguous()):
        return
    plans = own.get("step_plans")
    key = _step_key(dtype, p)
    if plans is not None and key in plans:
        return
    fwd, bwd = _ops.launched(plan, "fwd", dtype, p), _ops.launched(plan, "bwd", dtype, p)       # (notes of THIS operand type and width only)
    if fwd is None or bwd is None or fwd[0] != bwd[0] or min(fwd[2], bwd[2]) < SETTLE_AFTER:
        return                              # (forward and backward on different families, or a choice that is still changing)
    family = fwd[0]
    flat = plan.core.flat if batched else plan
    cores = [plan.core] + ([plan.core.t.core] if plan.core.t is not None else [])
    if batched and flat is not None:
        cores += [flat.core] + ([flat.core.t.core] if flat.core.t is not None else [])
    if any(not f.done() for c in cores for f in list(c.pending.values())):
        return                              # (a plan is still being built: the choice may change when it arrives)
    dev = plan.crow.device
    vt = _be._VTYPE[dtype]
    if family == "lattice":
        if flat is None:
            return
        (lpf, cf), (lps, cs, lpt, ct) = fwd[1], bwd[1]
        if _ops._lt.TUNE and not torch.are_deterministic_algorithms_enabled() and not all(
                getattr(c, "march", False) or c.tuned for c in (cf, cs, ct)):
            return                          # (a sweep configuration that is still to be measured: _ops._lattice_cfg)
        icrow, icol = (flat.crow, flat.col) if batched else _own_indices(flat)      # (the block-diagonal arrays are the cache's own)
        prods, tables = [], [icrow, icol]
        for mode, lp, cfg in ((_be.LAT_SPMM, lpf, cf), (_be.LAT_SDDMM, lps, cs), (_be.LAT_SPMMT, lpt, ct)):
            blob = ctypes.string_at(cfg.struct_addr, ctypes.sizeof(cfg.struct))     # sizes + device pointers into the tables below
            if getattr(cfg, "march", False):
                if cfg.col_tile != p:
                    return                  # (operands wider than a column tile run as several launches: the Python path)
                prods.append((0, blob, int(mode == _be.LAT_SPMMT)))
            else:
                prods.append((1, blob, 0))
            tables += _tensors_of(lp) + _tensors_of(cfg)
        sp = _host.StepPlan(icrow, icol, flat.n_rows, flat.n_cols, flat.nnz, p, vt, dev.index, prods[0], prods[1], prods[2], tables)
        if batched:
            sp.set_batch(plan.batch, plan.n_rows, plan.n_cols, plan.nnz)
    elif family == "tiles":
        # forward and SDDMM on the stored pattern's plan, Aᵀ·G on the transposed pattern's (its chunks read A's own values); batched
        # operands: the plans of the block-diagonal problem, the tensors keep their batch shape
        (tpf,), (tp, tt) = fwd[1], bwd[1]
        if tpf is not tp or flat is None:
            return
        blob = lambda q: ctypes.string_at(_be._tile_struct(q), ctypes.sizeof(_TilePlanStruct))      # noqa: E731
        icrow, icol = (flat.crow, flat.col) if batched else _own_indices(flat)
        sp = _host.StepPlan(icrow, icol, flat.n_rows, flat.n_cols, flat.nnz, p, vt, dev.index,
                            (3, blob(tp), 0), (3, blob(tp), 0), (3, blob(tt), 1), [icrow, icol] + _tensors_of(tp) + _tensors_of(tt))
        if batched:
            sp.set_batch(plan.batch, plan.n_rows, plan.n_cols, plan.nnz)
    elif family == "plan-free":
        # the step is on the plan-free kernels for good only once every structured plan has been asked for and has not come: the
        # pattern has to come back a few times (its row-pair / tile plans are requested on the way, _ops.PLAN_AFTER_USES)
        if dtype not in (torch.float32, torch.bfloat16, torch.float64) or fwd[2] <= _ops.PLAN_AFTER_USES + 2:
            return
        (t,) = bwd[1]
        if not (plan.crow.is_contiguous() and plan.col.is_contiguous() and t.crow.is_contiguous() and t.col.is_contiguous()
                and t.perm is not None and t.perm.is_contiguous()):
            return
        none = (2, b"", 0)
        b = plan.batch or 1
        icrow, icol = _own_indices(plan)                # (the plan-free kernels READ these arrays: same content as the caller's)
        sp = _host.StepPlan(icrow, icol, b * plan.n_rows, b * plan.n_cols, b * plan.nnz, p, vt, dev.index, none, none, none, [])
        sp.set_plan_free(_be.itype_of(plan.crow), t.crow, t.col, t.perm, int(plan.max_row_nnz), int(t.max_row_nnz),
                         bool(_be.fused_backward_supported(dtype, p)))
        if batched:
            # (batched operands off a lattice — the reference's own batched benchmark shape, 128 items of 1024 x 1024 with 4096 random
            # entries, benchmarks/results/batched_sparse_mm_rand_results.csv:31: 0.07 ms of kernels per step against 0.24-0.30 ms of Python;
            # the kernels take the batch as it is: item strides, per-item transposed pattern)
            sp.set_batch(plan.batch, plan.n_rows, plan.n_cols, plan.nnz)
    else:
        return                              # (row pairs: the Python path)
    if op.layout != torch.sparse_csr:
        sp.set_coo(op.indices)
    if plans is None:
        plans = own["step_plans"] = {}
    plans[key] = sp


def _own_indices(g) -> tuple:
    """(crow, col) of the RowGather `g` as tensors the pattern cache OWNS (its index copy, or a copy made here once): a StepPlan lives in
    the cache entry, and an entry that held the caller's index tensors would keep them — and itself — alive for ever."""
    own = g.core.own
    got = own.get("step_indices")
    if got is None:
        copy = own.get("index_copy")
        if (copy is not None and len(copy) == 2 and copy[0].shape == g.crow.shape and copy[1].shape == g.col.shape
                and copy[0].dtype == g.crow.dtype and copy[1].dtype == g.col.dtype):
            got = copy                                   # (the CSR tensors the entry was built from, copied by the fingerprint pass …
            _pt._after_its_writer(g.core, torch.cuda.current_stream(g.crow.device))     # … possibly on another stream: ordered behind it)
        else:
            got = (g.crow.contiguous().clone(), g.col.contiguous().clone())
        own["step_indices"] = got
    return got


def _tensors_of(obj, depth: int = 2):
    """Every tensor reachable from the attributes of a plan / configuration object (its device tables), `depth` levels deep."""
    found = []
    names = getattr(type(obj), "__slots__", None) or list(getattr(obj, "__dict__", {}))
    for k in names:
        v = getattr(obj, k, None)
        if torch.is_tensor(v):
            found.append(v)
        elif depth > 1 and v is not None and not isinstance(v, (int, float, str, bytes, bool, tuple, list, dict, ctypes.Structure)):
            found += _tensors_of(v, depth - 1)
        elif isinstance(v, (tuple, list)):
            found += [t for t in v if torch.is_tensor(t)]
    return found


class _Operand:
    """The sparse operand as the kernels see it: a row-gather plan + the value array, plus how
    to hand a value-shaped gradient back in the caller's layout."""

    __slots__ = ("plan", "values", "layout", "indices", "shape", "flat_batch")

    def __init__(self, A: torch.Tensor):
        self.layout = A.layout
        self.shape = A.shape
        self.flat_batch = None
        if A.layout == torch.sparse_csr:
            self.plan = _pt.from_csr(A)
            self.values = A.values()
            self.indices = None
            return
        # COO.  Batched COO is flattened to one block-diagonal 2-D pattern (what the reference
        # does for every batched input, sparse_matmul.py:151-153); items may differ in nnz.
        if A.dim() == 3:
            A = A if A.is_coalesced() else A.coalesce()
            idx = A._indices()
            self.indices = idx
            self.flat_batch = A.size(0)
            self.plan = _pt.from_coo_batched(idx, A.shape)
            self.values = A._values()
            return
        self.indices = A._indices()
        self.values = A._values()
        self.plan = _pt.from_coo_2d(self.indices, A.shape, coalesced=A.is_coalesced())

    def rebuild(self, grad_values: torch.Tensor) -> torch.Tensor:
        """Sparse gradient with A's own layout/indices (reference sparse_matmul.py:208-219)."""
        if self.layout == torch.sparse_csr:
            return torch.sparse_csr_tensor(self.plan.crow, self.plan.col, grad_values, self.shape)
        return torch.sparse_coo_tensor(self.indices, grad_values, self.shape)


class SparseMatMul(torch.autograd.Function):
    """Autograd kernel behind :func:`sparse_mm` (mirrors reference ``sparse_matmul.py:132-234``)."""

    @staticmethod
    def forward(ctx, A, B):
        ctx.batch_size = B.size()[0] if B.dim() == 3 else None
        ctx.A_shape = A.size()
        ctx.B_shape = B.size()
        grad_flag = A.requires_grad or B.requires_grad

        A, B = A.detach(), B.detach()
        if A.device != B.device:
            raise RuntimeError(f"A and B must be on the same device, got {A.device} and {B.device}")

        op = _Operand(A)
        plan = op.plan
        if op.flat_batch is not None:  # batched COO → block-diagonal 2-D problem
            Bk = B.reshape(-1, B.size(-1))
        else:
            Bk = B
        if A.dtype != B.dtype:
            raise RuntimeError(f"expected A and B to have the same dtype, got {A.dtype} and {B.dtype}")
        x = _ops.spmm(plan, op.values, Bk)
        if op.flat_batch is not None:
            x = x.view(ctx.batch_size, ctx.A_shape[-2], ctx.B_shape[-1])

        ctx.op = op
        ctx.save_for_backward(op.values, Bk)
        x.requires_grad_(grad_flag)
        return x

    @staticmethod
    def backward(ctx, grad):  # type: ignore[override]
        values, B = ctx.saved_tensors
        op: _Operand = ctx.op
        plan = op.plan
        gradA = gradB = None

        G = grad.reshape(-1, grad.size(-1)) if op.flat_batch is not None else grad

        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if (need_a and need_b and plan.perm is None and G.dtype == B.dtype == values.dtype
                and _be.fused_backward_supported(G.dtype, G.size(-1))):
            # both gradients in one pass: each upstream row G[i,:] is gathered once (reference :172-229)
            gvals, gradB = _ops.mm_backward(plan, values, G, B)
            gradA = op.rebuild(gvals)
            if ctx.batch_size is not None:
                gradB = gradB.view(ctx.B_shape)
            if op.flat_batch is None:
                _settle_step_plan(op, values, G, B)
            return gradA, gradB

        if need_a and need_b and plan.perm is None and G.dtype == B.dtype == values.dtype:
            # both gradients as two products (operand types the fused walk is not compiled for: fp64, very wide rows)
            gvals, gradB = _ops.mm_backward_separate(plan, values, G, B)
            gradA = op.rebuild(gvals)
            if ctx.batch_size is not None:
                gradB = gradB.view(ctx.B_shape)
            if op.flat_batch is None:
                _settle_step_plan(op, values, G, B)
            return gradA, gradB

        if need_a:
            # gradA[k] = <G[row k,:], B[col k,:]> at A's stored entries only (reference :172-205)
            if plan.perm is None:
                gvals = _ops.sddmm(plan, G, B)
            else:  # un-coalesced COO: one gradient entry per stored duplicate, in A's own order
                gvals = _be.coo_sddmm(op.indices[0], op.indices[1], G, B)
            gradA = op.rebuild(gvals)

        if need_b:
            # gradB = Aᵀ·G as a gather over the cached transposed pattern (reference :229)
            gradB = _ops.spmm_t(plan, values, G)
            if ctx.batch_size is not None:
                gradB = gradB.view(ctx.B_shape)

        return gradA, gradB
