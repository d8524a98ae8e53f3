"""Kernel selection for one sparse operand: row-pair union kernels when neighbouring rows share columns (stencils,
banded factors, meshes) and the operands qualify, plain gather kernels otherwise.  Both produce the same values
(same per-row summation order when a pair has one entry lane, except that brick plans of 3-D lattices sum a row of the
transposed product plane by plane: equal to rounding); the choice is speed only.

Batched CSR operands (torch layout, equal nnz per item) that qualify for the row-pair kernels are handed to them as
ONE block-diagonal 2-D problem (`_pattern.flat_of`: two vectorised adds on the index arrays, values untouched):
the row-pair plans are translation-deduplicated, so items that share a pattern also share their plan records."""

from __future__ import annotations

import os
import threading

import torch

from . import _backend as _be
from . import _cpu
from . import _lattice as _lt
from . import _pattern as _pt
from ._pattern import RowGather

# Row-pair kernels (csrc/rowpack_impl.h): rows 2q, 2q+1 walk the union of their columns, a shared dense row is
# gathered once.  First choice for forward, transposed and fused-backward walks when neighbouring rows share columns
# (stencil / banded / mesh patterns); TSGU_ENABLE_PACK=0 disables.  Small patterns stay on the plan-free kernels
# (launch-bound anyway).
ENABLE_PACK = os.environ.get("TSGU_ENABLE_PACK", "1") == "1"
PACK_MIN_NNZ = 1 << 16
# A row-pair plan costs a few device sorts: it is built when a pattern is seen for the PLAN_AFTER_USES-th time (the
# first use runs on the plan-free kernels), so that one-off patterns never pay for it.  0 = build at first sight.
PLAN_AFTER_USES = int(os.environ.get("TSGU_PLAN_AFTER_USES", "1"))
# ... and it is built on a worker thread + side stream (TSGU_PLAN_ASYNC=0: inline): the steps in between keep running
# on the plan-free kernels, no step ever waits for a plan.  `torchsparsegradutils_amd.wait_for_plans()` joins.
PLAN_ASYNC = os.environ.get("TSGU_PLAN_ASYNC", "1") == "1"


# Row-block tile kernels (csrc/tile_impl.h): general patterns whose neighbouring rows share columns (mesh orderings, banded factors,
# FEM matrices): a block's distinct dense rows are staged in LDS one block ahead of the walk.  Chosen before the row pairs when the
# pattern qualifies (every block's tile fits, entries share dense rows); TSGU_ENABLE_TILE=0 disables.
ENABLE_TILE = os.environ.get("TSGU_ENABLE_TILE", "1") == "1"
# Operands wider than one column tile (32 fp32 columns) run in ONE launch (round 6: the pipeline's steps are (block, column tile) pairs,
# a block's values / entry bytes / row pointers are staged once).  Round 5 ran one launch per column tile and kept wide operands over
# fewer than 8192 blocks on the plan-free kernels; TSGU_TILE_WIDE_MIN_BLOCKS restores such a threshold for A/B measurements.
TILE_COLUMNS = 32
TILE_WIDE_MIN_BLOCKS = int(os.environ.get("TSGU_TILE_WIDE_MIN_BLOCKS", "0"))


# Lattice plane-sweep kernels (csrc/lattice_impl.h): patterns that are stencils on a row-major lattice (what the
# reference's PairwiseEncoder and its stencil benchmarks produce) are walked tile by tile with the halo of the dense
# operand in LDS.  First choice when the pattern qualifies; anything else takes the row-pair / plan-free kernels.
ENABLE_LATTICE = os.environ.get("TSGU_ENABLE_LATTICE", "1") == "1"
# value types the sweeps are compiled for (fp32 accumulation for bf16)
LATTICE_DTYPES = (torch.float32, torch.bfloat16, torch.float64)
# measured launch configurations: every trial launch follows a 256 MB device copy (the cache state of a step, not of a back-to-back loop)
TUNE_COLD = os.environ.get("TSGU_TUNE_COLD", "1") != "0"


# ---- what a step launched ----------------------------------------------------------------------------------------------------------
# The order of preference between the kernel families (lattice -> batched row pairs -> row-block tiles -> row pairs -> plan-free) is
# written down ONCE, in spmm() / mm_backward() below.  The step's C++ host path (sparse_matmul._settle_step_plan) does not decide
# anything a second time: the entry points NOTE what they launched — family and the plan objects — with the pattern, and the host
# path is derived from notes that have stopped changing.  (Round 5 re-derived the decision there and a finished-but-unclaimed
# row-pair future kept every tile pattern off the C++ path.)
_CHOICE = threading.local()


def _chose(family: str, *payload) -> None:
    """What the product that is about to be launched runs on (read by mm_backward_separate right after the call, same thread)."""
    _CHOICE.last = (family, payload)


def _note(plan: RowGather, product: str, dense: torch.Tensor, family: str, *payload) -> None:
    own = plan.core.own
    sig = (dense.dtype, dense.size(-1))          # a pattern may be used with operands of several types and widths: a note is about ONE
    prev = own.get("launched_" + product)
    same = (prev is not None and prev[0] == family and prev[3] == sig and len(prev[1]) == len(payload)
            and all(a is b for a, b in zip(prev[1], payload)))
    own["launched_" + product] = (family, payload, (prev[2] + 1) if same else 1, sig)


def launched(plan: RowGather, product: str, dtype=None, p: int = 0):
    """(family, payload, consecutive steps with this very choice, (dtype, p)) of the last `product` ("fwd" / "bwd") on the pattern —
    None when there is none or (dtype given) when it was made with operands of another type or width."""
    got = plan.core.own.get("launched_" + product)
    if got is not None and dtype is not None and got[3] != (dtype, p):
        return None
    return got


def _lattice_plan(plan: RowGather, transposed: bool = False):
    """LatticePlan of the stored-order walk of the 2-D `plan` (or, `transposed`, of the walk of its transposed pattern; the
    values stay in `plan`'s stored order), None when the pattern is not a lattice stencil.  Cached with the pattern.
    Built at FIRST sight: the row kernels of csrc/lattice_plan.hip make it a few milliseconds (two passes over the
    pattern + a sort of one word per row), and the transposed walk needs no transposed pattern."""
    if (not ENABLE_LATTICE or not plan.crow.is_cuda or plan.batch is not None or plan.perm is not None or plan.nnz < PACK_MIN_NNZ
            or plan.n_rows != plan.n_cols):
        return None          # (plans exist for GPU operands only: CPU operands take _cpu.py)
    own = plan.core.own
    # a plan is built with host round trips (status words, class tables): never inside a stream capture — a pattern first
    # seen there runs on the plan-free kernels and gets its plan from the first call outside the capture
    capturing = None
    if "lattice" not in own:
        # index tensors of this geometry keep arriving with new content (_pattern._core_for): no analysis until THIS pattern has
        # come back for a third step — one-off patterns run plan-free (0.7 ms at C2) instead of paying ~11 ms of plan each
        sightings = own.get("volatile")
        if sightings is not None and sightings < 6:
            own["volatile"] = sightings + 1
            return None
        capturing = plan.crow.is_cuda and torch.cuda.is_current_stream_capturing()
        if capturing:
            return None
        own["lattice"] = _lt.build_lattice_plan_hip(plan, _be)
    fwd = own["lattice"]
    if not transposed or fwd is None:
        return fwd
    if "lattice_t" not in own:
        if capturing is None:
            capturing = plan.crow.is_cuda and torch.cuda.is_current_stream_capturing()
        if capturing:
            return None
        own["lattice_t"] = _lt.build_lattice_plan_hip(plan, _be, forward=fwd)
    return own["lattice_t"]


def _lattice_cfg(plan: RowGather, mode: int, dense: torch.Tensor, *others: torch.Tensor):
    """(LatticePlan, LatticeConfig) for these operands or None; `plan` is always the pattern the values are stored in (the
    transposed product walks its transposed pattern through the plan's own arrays)."""
    if not ENABLE_LATTICE:
        return None
    # steady state: contiguous 16-byte aligned operands of a pattern whose configuration is final — one dictionary lookup
    # (the checks below cost the host more than the launch itself, and a step asks three times)
    plain = dense.dim() == 2 and dense.is_contiguous() and dense.data_ptr() % 16 == 0
    for t in others:
        plain = plain and t.dim() == 2 and t.is_contiguous() and t.data_ptr() % 16 == 0
    memo = key = None
    if plain and plan.batch is None and plan.perm is None:
        memo = plan.core.own.get("lattice_memo")
        if memo is None:
            memo = plan.core.own["lattice_memo"] = {}
        key = (mode, dense.dtype, dense.size(-1), ENABLE_LATTICE, _lt.ENABLE_MARCH)
        hit = memo.get(key)
        if hit is not None:
            return hit
    if dense.dim() != 2 or not _be._tiled_ok(*(_be.rowmajor(t) for t in (dense,) + others)):
        return None
    if dense.dtype not in LATTICE_DTYPES:
        return None
    row_bytes = dense.size(-1) * dense.element_size()
    lanes = row_bytes // 16
    wide = dense.dtype == torch.float32 and dense.size(-1) > 64 and dense.size(-1) % 64 == 0     # plane march only: column tiles of 64
    if not wide and (row_bytes % 16 or lanes not in (1, 2, 4, 8, 16) or (lanes == 1 and not (mode == _be.LAT_SPMM and dense.dtype == torch.float32))):
        return None            # dense rows the sweeps are not compiled for: do not even analyse the pattern
    fwd = _lattice_plan(plan)
    if fwd is None:
        return None
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing and fwd._march is False:
        return None            # (the march tables are copied to the device when they are first derived: not inside a capture)
    # box stencils (periodic or truncated; 27-point, 7-point, triangular parts …): the plane-march kernels — all three products
    # from the stored-order plan alone
    cfg = _be.march_config(fwd, mode, dense.dtype, dense.size(-1))
    if cfg is not None:
        if memo is not None:
            memo[key] = (fwd, cfg)
        return fwd, cfg
    if wide:
        return None            # (no general-sweep form of these)
    lp = _lattice_plan(plan, transposed=True) if mode == _be.LAT_SPMMT else fwd
    if lp is None:
        return None
    if capturing and (mode, _be._VTYPE[dense.dtype], dense.size(-1)) not in lp._cfg:
        return None            # (record tables and class lists of a configuration are built with host round trips)
    cfg = _be.lattice_config(lp, mode, dense.dtype, dense.size(-1))
    if cfg is None:
        return None
    # (deterministic mode: the launch configuration — and with it the grouping of the Krylov loops' per-workgroup dot partials —
    # must not depend on a wall-clock trial; the ranked configuration stays.  TSGU_LATTICE_TUNE=0 does the same for a process)
    if _lt.TUNE and not cfg.tuned and not torch.are_deterministic_algorithms_enabled():
        cfg.uses += 1
        # (never beside a plan build on the worker's side stream: its device sorts would be in the trial timings)
        if cfg.uses >= _lt.TUNE_AFTER_USES and not torch.cuda.is_current_stream_capturing() and not _pt.plans_in_flight():
            try:
                cfg = _measured_cfg(lp, mode, dense, cfg)
            except torch.cuda.OutOfMemoryError:
                cfg.tuned = True      # no room for the trial operands: the ranked configuration stays
    if memo is not None and (cfg.tuned or not _lt.TUNE) and not torch.are_deterministic_algorithms_enabled():
        memo[key] = (lp, cfg)
    return lp, cfg


def _measured_cfg(lp, mode: int, dense: torch.Tensor, cfg):
    """The pattern keeps coming back: time the best-ranked launch configurations once on operands of the caller's shape
    (`_lattice.tune_config`; every configuration gives the same bits) and keep the fastest."""
    val = torch.zeros(lp.nnz, dtype=dense.dtype, device=dense.device)

    def run(c):
        if mode == _be.LAT_SDDMM:
            _be.csr_sddmm_lattice(lp, c, dense, dense)
        else:
            _be.csr_spmm_lattice(lp, c, val, dense)

    # Between two launches of one of these kernels a step (or a solver iteration) streams several hundred MB through the chip: what
    # the kernel finds in L2 / MALL is NOT what its own previous launch left there.  Back-to-back timings hide the difference between
    # configurations (C4's K1: 512 and 1024 threads time the same back to back, 26.6 against 31.4 us inside the CG loop), so every
    # timed launch follows a 256 MB device copy, and is timed by its own event pair.
    evict = None
    if TUNE_COLD:
        try:
            evict = (torch.empty(64 << 20, dtype=torch.float32, device=dense.device), torch.empty(64 << 20, dtype=torch.float32, device=dense.device))
        except torch.cuda.OutOfMemoryError:
            evict = None

    def time_ms(c):
        run(c)
        if evict is None:
            # the best of three timings of six launches: a single timing of four picked a 30 % slower configuration now and then
            # (clock ramps, a neighbour's launch) and the choice is final for the pattern
            best = None
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(6):
                    run(c)
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1) / 6
                best = t if best is None or t < best else best
            return best
        pairs = []
        for _ in range(9):
            evict[1].copy_(evict[0])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(c)
            e1.record()
            pairs.append((e0, e1))
        pairs[-1][1].synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in pairs)
        return sum(ts[1:5]) / 4          # (the mean of the second to fifth fastest of nine: no ramp-up launch, no neighbour's spike)

    events, _be.KERNEL_EVENTS = _be.KERNEL_EVENTS, None      # (bench.py's per-kernel hook does not see the trial launches)
    try:
        with torch.cuda.device(dense.device):
            return _be.lattice_tune(lp, mode, dense.dtype, dense.size(-1), time_ms) or cfg
    finally:
        _be.KERNEL_EVENTS = events


def _pack_for(plan: RowGather, dense: torch.Tensor, *others: torch.Tensor, need_plain_slots: bool = False):
    """RowPackPlan of the 2-D `plan` for these dense operands, or None (disabled / not supported / not profitable /
    pattern not seen often enough yet)."""
    if (not ENABLE_PACK or not plan.crow.is_cuda or not dense.is_cuda or plan.batch is not None or dense.dim() != 2
            or plan.nnz < PACK_MIN_NNZ or plan.crow.dtype not in (torch.int32, torch.int64)):
        return None
    geo = _be.rowpack_geometry(dense.dtype, dense.size(-1))
    if geo is None or not _be._tiled_ok(*(_be.rowmajor(t) for t in (dense,) + others)):
        return None
    rpb, limits, entry_lanes = geo
    if need_plain_slots and entry_lanes > 1:
        return None  # SDDMM walks ownership-bit records (one entry lane per pair)
    if not plan.seen_enough(PLAN_AFTER_USES):
        return None
    # (deterministic mode: the step at which the kernels switch must not depend on a worker thread's timing)
    if PLAN_ASYNC and PLAN_AFTER_USES > 0 and not torch.are_deterministic_algorithms_enabled():
        return plan.rowpack_plan_async(rpb, limits, explicit_slots=entry_lanes > 1)
    return plan.rowpack_plan(rpb, limits, explicit_slots=entry_lanes > 1)


def _tile_for(plan: RowGather, dense: torch.Tensor, *others: torch.Tensor):
    """TilePlan of the 2-D `plan` for these dense operands, or None (disabled / not compiled for them / the pattern does not qualify
    or has not come back yet)."""
    if (not ENABLE_TILE or not plan.crow.is_cuda or not dense.is_cuda or plan.batch is not None or dense.dim() != 2
            or plan.nnz < PACK_MIN_NNZ or plan.crow.dtype not in (torch.int32, torch.int64)):
        return None
    geo = _be.tile_geometry(dense.dtype, dense.size(-1))
    if geo is None:
        return None
    ops = tuple(_be.rowmajor(t) for t in (dense,) + others)
    if not _be._tiled_ok(*ops) or any(t.dtype != dense.dtype or t.size(0) * max(t.stride(0), 1) * t.element_size() >= 2**32 for t in ops):
        return None
    if max(plan.n_cols, plan.n_rows) >= 1 << 24 or any(t.stride(0) * t.element_size() >= 1 << 24 for t in ops):
        return None          # (the kernels' tile row offsets are 24-bit products)
    if dense.size(-1) > TILE_COLUMNS and (plan.n_rows + geo[0] - 1) // geo[0] < TILE_WIDE_MIN_BLOCKS:
        return None          # (several column tiles over few blocks: see TILE_WIDE_MIN_BLOCKS)
    if not plan.seen_enough(PLAN_AFTER_USES):
        return None
    key = ("tile",) + tuple(geo)
    if key in plan.core.packs:
        return plan.core.packs[key]
    if torch.cuda.is_current_stream_capturing():
        return None          # (a plan is built with host reads: never inside a stream capture — asked for again after it)
    if PLAN_ASYNC and PLAN_AFTER_USES > 0 and not torch.are_deterministic_algorithms_enabled():
        return plan.tile_plan(geo, asynchronous=True)
    return plan.tile_plan(geo)


def _flat(plan: RowGather, *dense: torch.Tensor):
    """(block-diagonal 2-D plan, flattened dense operands) of a batched problem, or None when the operands are not
    batch-contiguous (the plain kernels then take the batch as gridDim.y)."""
    if plan.batch is None or any(t.dim() != 3 for t in dense):
        return None
    flat = []
    for t in dense:
        t = _be.rowmajor(t)
        if t.size(0) > 1 and t.stride(0) != t.size(1) * t.stride(1):
            return None
        if t.size(1) > 1 and t.stride(1) != t.size(2):
            return None
        flat.append(t.reshape(-1, t.size(-1)))
    return _pt.flat_of(plan), flat


def _lattice_backward(plan: RowGather, values: torch.Tensor, G: torch.Tensor, B: torch.Tensor):
    """Both gradients of C = A·B by two plane sweeps: the SDDMM in A's stored order (gradA leaves fully coalesced) and
    Aᵀ·G on the transposed pattern with the values read from A's own array.  None when the pattern is not a lattice."""
    if plan.perm is not None:
        return None
    fplan, Gf, Bf = plan, G, B
    if plan.batch is not None:
        fl = _flat(plan, G, B)
        if fl is None:
            return None
        fplan, (Gf, Bf) = fl
    vals = values.reshape(-1)
    fwd = _lattice_cfg(fplan, _be.LAT_SDDMM, Bf, Gf)
    if fwd is None:
        return None
    bwd = _lattice_cfg(fplan, _be.LAT_SPMMT, Gf)
    if bwd is None:
        return None
    ga = _be.csr_sddmm_lattice(fwd[0], fwd[1], Gf, Bf)
    gb = _be.csr_spmm_lattice(bwd[0], bwd[1], vals, Gf)
    _note(plan, "bwd", G, "lattice", fwd[0], fwd[1], bwd[0], bwd[1])
    return ga.view(values.shape), gb.view(B.shape)


def mm_backward(plan: RowGather, values: torch.Tensor, G: torch.Tensor, B: torch.Tensor):
    """(gradA values in A's order, gradB) of C = A·B in one pass over the transposed pattern."""
    if not G.is_cuda:       # CPU operands: the torch-op path (_cpu.py), chosen by the operands' device and nothing else
        return _cpu.sddmm(plan, G, B), _cpu.spmm(plan.transposed, values, G)
    same = values.dtype == G.dtype == B.dtype
    if same and ENABLE_LATTICE and values.dtype in LATTICE_DTYPES:
        got = _lattice_backward(plan, values, G, B)
        if got is not None:
            return got
    if same and plan.batch is not None and plan.perm is None and ENABLE_TILE and _be.tile_geometry(B.dtype, B.size(-1)) is not None:
        # batched operands whose items are meshes: the block-diagonal problem on the row-block tiles (round 6; the tile plan of a
        # block-diagonal pattern is the items' plans one after the other — a block of 64 rows never spans two items' columns unless
        # n is not a multiple of 64, and then its tile simply lists both)
        fl = _flat(plan, G, B)
        if fl is not None:
            fplan, (Gf, Bf) = fl
            tp, tt = _tile_for(fplan, Bf, Gf), _tile_for(fplan.transposed, Gf)
            if tp is not None and tt is not None:
                _note(plan, "bwd", G, "tiles", tp, tt)
                ga, gb = _be.csr_sddmm_tile(tp, Gf, Bf), _be.csr_spmm_tile(tt, values.reshape(-1), Gf)
                return ga.view(values.shape), gb.view(B.shape)
    if same and plan.batch is not None and ENABLE_PACK:
        fl = _flat(plan, G, B)
        if fl is not None:
            fplan, (Gf, Bf) = fl
            rp = _pack_for(fplan.transposed, Gf, Bf)
            if rp is not None:
                ga, gb = _be.csr_mm_backward_rowpack(fplan.transposed.crow, rp, values.reshape(-1), Gf, Bf, fplan.n_cols)
                _note(plan, "bwd", G, "row pairs", rp)
                return ga.view(values.shape), gb.view(B.shape)
    t = plan.transposed
    if same and plan.batch is None and plan.perm is None:
        # row-block tiles: the SDDMM in stored order + the transposed product on the transposed pattern's tiles (A's own values)
        tp, tt = _tile_for(plan, B, G), _tile_for(t, G)
        if tp is not None and tt is not None:
            _note(plan, "bwd", G, "tiles", tp, tt)
            return _be.csr_sddmm_tile(tp, G, B), _be.csr_spmm_tile(tt, values, G)
    rp = _pack_for(t, G, B) if same else None
    if rp is not None and rp.srcstart is not None and plan.batch is None and plan.perm is None:
        # the transposed plan reached the dictionary form through row-relative value positions (mesh orderings): the SDDMM on the
        # stored-order plan + the transposed product beat the fused walk (mesh27_blocked: 124 + 208 us against 417 us)
        _note(plan, "bwd", G, "row pairs", rp)
        return sddmm(plan, G, B), spmm(t, values, G, owner=plan)
    if rp is not None:
        _note(plan, "bwd", G, "row pairs", rp)
        return _be.csr_mm_backward_rowpack(t.crow, rp, values, G, B, t.n_rows)
    _note(plan, "bwd", G, "plan-free", t)
    return _be.csr_mm_backward(t, values, G, B, plan.n_rows, plan.n_cols)


def spmm(plan: RowGather, values: torch.Tensor, B: torch.Tensor, owner: RowGather = None) -> torch.Tensor:
    """A·B for the operand described by (plan, values); perm-aware (transposed / un-coalesced plans).  `owner`: when
    `plan` is `owner.transposed`, the pattern whose stored order the values are in (lets Aᵀ·G take the lattice sweep)."""
    if not B.is_cuda:
        return _cpu.spmm(plan, values, B)
    if values.dtype == B.dtype and not _be.is_transposed_view(B):  # transposed views: zero-copy column-strided K1
        stored = plan.perm is None
        if ENABLE_LATTICE and (stored or (owner is not None and owner.perm is None)):
            src = plan if stored else owner
            fsrc, Bf = src, B
            if src.batch is not None:
                fl = _flat(src, B)
                fsrc, Bf = (fl[0], fl[1][0]) if fl is not None else (None, None)
            got = None
            if fsrc is not None:
                got = _lattice_cfg(fsrc, _be.LAT_SPMM if stored else _be.LAT_SPMMT, Bf)
            if got is not None:
                out = _be.csr_spmm_lattice(got[0], got[1], values.reshape(-1), Bf)
                _chose("lattice", got[0], got[1])
                if stored:
                    _note(plan, "fwd", B, "lattice", got[0], got[1])
                return out.view(B.shape[:-2] + (plan.n_rows, B.size(-1)))
        if plan.batch is not None and stored and ENABLE_TILE and _be.tile_geometry(B.dtype, B.size(-1)) is not None:
            fl = _flat(plan, B)
            if fl is not None:
                fplan, (Bf,) = fl
                tp = _tile_for(fplan, Bf)
                if tp is not None:
                    _chose("tiles", tp)
                    _note(plan, "fwd", B, "tiles", tp)
                    return _be.csr_spmm_tile(tp, values.reshape(-1), Bf).view(B.size(0), plan.n_rows, B.size(-1))
        if plan.batch is not None and ENABLE_PACK:
            fl = _flat(plan, B)
            if fl is not None:
                fplan, (Bf,) = fl
                rp = _pack_for(fplan, Bf)
                if rp is not None:
                    out = _be.csr_spmm_rowpack(fplan.crow, values.reshape(-1), rp, Bf, fplan.n_rows)
                    _chose("row pairs", rp)
                    _note(plan, "fwd", B, "row pairs", rp)
                    return out.view(B.size(0), plan.n_rows, B.size(-1))
        tp = _tile_for(plan, B)
        if tp is not None:
            _chose("tiles", tp)
            _note(plan, "fwd", B, "tiles", tp)
            return _be.csr_spmm_tile(tp, values, B)
        rp = _pack_for(plan, B)
        if rp is not None:
            _chose("row pairs", rp)
            _note(plan, "fwd", B, "row pairs", rp)
            return _be.csr_spmm_rowpack(plan.crow, values, rp, B, plan.n_rows)
    _chose("plan-free", plan)
    _note(plan, "fwd", B, "plan-free")
    return _be.csr_spmm(plan.crow, plan.col, values, B, plan.n_rows, plan.n_cols, perm=plan.perm, max_row_nnz=plan.max_row_nnz)


def spmm_t(owner: RowGather, values: torch.Tensor, G: torch.Tensor) -> torch.Tensor:
    """Aᵀ·G for the operand (owner, values): the lattice sweep walks the transposed pattern through the owner's own arrays;
    everything else goes through the cached transposed pattern (built on first use)."""
    if not G.is_cuda:
        return _cpu.spmm(owner.transposed, values, G)
    if ENABLE_LATTICE and owner.perm is None and values.dtype == G.dtype and not _be.is_transposed_view(G):
        fsrc, Gf = owner, G
        if owner.batch is not None:
            fl = _flat(owner, G)
            fsrc, Gf = (fl[0], fl[1][0]) if fl is not None else (None, None)
        got = _lattice_cfg(fsrc, _be.LAT_SPMMT, Gf) if fsrc is not None else None
        if got is not None:
            out = _be.csr_spmm_lattice(got[0], got[1], values.reshape(-1), Gf)
            _chose("lattice", got[0], got[1])
            return out.view(G.shape[:-2] + (owner.n_cols, G.size(-1)))
    return spmm(owner.transposed, values, G, owner=owner)


def sddmm(plan: RowGather, G: torch.Tensor, B: torch.Tensor, alpha: float = 1.0, swap_roles: bool = False) -> torch.Tensor:
    """alpha·<G[row k], B[col k]> (or roles swapped) at the plan's stored entries, in plan order."""
    if not G.is_cuda:
        return _cpu.sddmm(plan, G, B, alpha=alpha, swap_roles=swap_roles)
    gathered = G if swap_roles else B
    rowop = B if swap_roles else G
    if plan.perm is None and G.dtype == B.dtype:
        got = _lattice_cfg(plan, _be.LAT_SDDMM, gathered, rowop)
        if got is not None:
            _chose("lattice", got[0], got[1])
            return _be.csr_sddmm_lattice(got[0], got[1], rowop, gathered, alpha=alpha)
        tp = _tile_for(plan, gathered, rowop)
        if tp is not None:
            _chose("tiles", tp)
            return _be.csr_sddmm_tile(tp, rowop, gathered, alpha=alpha)
        rp = _pack_for(plan, gathered, rowop, need_plain_slots=True)
        if rp is not None and rp.upos is None:
            _chose("row pairs", rp)
            return _be.csr_sddmm_rowpack(plan.crow, rp, rowop, gathered, plan.n_rows, alpha=alpha)
    _chose("plan-free")
    return _be.csr_sddmm(plan.crow, plan.col, G, B, plan.n_rows, plan.n_cols, alpha=alpha, swap_roles=swap_roles)


def mm_backward_separate(plan: RowGather, values: torch.Tensor, G: torch.Tensor, B: torch.Tensor):
    """(gradA values in A's order, gradB) as TWO products — the SDDMM in stored order and Aᵀ·G — for operands the fused walk is not
    compiled for (fp64, very wide rows).  Noted like mm_backward when both products ran on the same family."""
    ga = sddmm(plan, G, B)
    a = getattr(_CHOICE, "last", None)
    gb = spmm_t(plan, values, G)
    b = getattr(_CHOICE, "last", None)
    if G.is_cuda and a is not None and b is not None and a[0] == b[0]:
        _note(plan, "bwd", G, a[0], *(a[1] + b[1]))
    return ga, gb
