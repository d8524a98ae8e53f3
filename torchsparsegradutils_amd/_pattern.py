"""Sparsity-pattern plans: the row-gather view of a sparse operand plus cached analyses.

The reference re-derives structure on every call (``repeat_interleave`` of the row pointer,
``A.t()`` → CSC→CSR conversion + sort inside ``torch.sparse.mm``; sparse_matmul.py:186-192,229).
Here the structure work is done once per sparsity pattern and cached on the identity of the
index tensors (the common training loop updates values, not the pattern):

* ``RowGather``   – (crow, col[, perm]) arrays the HIP kernels walk, for CSR / COO / batched inputs
* ``.transposed`` – the same for Aᵀ (CSC of A) with a permutation into A's value array, so that
  Aᵀ·G and transposed triangular solves are gather kernels (no atomics, deterministic)
* ``.has_diagonal`` – whether any stored entry sits on the diagonal (sparse_solve.py:230)
"""

from __future__ import annotations

import os as _os
import threading
import warnings
import weakref
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Tuple

import torch


_EXECUTOR = ThreadPoolExecutor(max_workers=1, thread_name_prefix="tsgu-plan")
_PENDING_LOCK = threading.Lock()
_PENDING_CORES = set()
_SIDE_STREAMS = {}


def _side_stream(dev: torch.device):
    s = _SIDE_STREAMS.get(dev)
    if s is None:
        s = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return s


def plans_in_flight() -> bool:
    """True while an asynchronous plan build is running (its device work shares the GPU with whatever the caller times)."""
    with _PENDING_LOCK:
        return any(not f.done() for c in _PENDING_CORES for f in c.pending.values())


def wait_for_plans() -> None:
    """Block until every asynchronous plan build submitted so far has finished (benchmarks call this at the end of
    their warm-up; a training loop never needs to).  The plans are picked up by the next call of the operator."""
    with _PENDING_LOCK:
        futs = [f for c in _PENDING_CORES for f in c.pending.values()]
    for f in futs:
        try:
            f.result()
        except Exception:  # noqa: BLE001  (reported by the operator that picks the result up)
            pass
    from . import _backend

    _backend.poll_errors(block=True)     # deferred device-side error words (lazy triangular-solve checks), if any


def _wait_quietly(fut) -> None:
    """Join an asynchronous plan build without re-raising what it raised."""
    try:
        fut.result()
    except Exception:  # noqa: BLE001
        pass


class _Core:
    """Everything derived from one sparsity pattern (shared by all RowGather views of it, owned by the cache).
    Holds only tensors this module allocated — never the caller's index tensors — so that dropping the sparse
    tensor releases its plans (the cache entry is evicted by a finalizer on the index storage)."""

    __slots__ = ("t", "has_diag", "rows", "packs", "pending", "uses", "flat", "own", "fp", "geom", "__weakref__")

    def __init__(self):
        self.fp = None      # [tensors][2] int64 on the device: content fingerprint of the index tensors (see _core_for), until read to the host
        self.geom = None    # (kind, shape, geometry of the index tensors): what a fingerprint is compared within
        self.t: Optional[RowGather] = None
        self.has_diag: Optional[bool] = None
        self.rows = None
        self.packs = {}
        self.pending = {}   # plan key -> Future of an asynchronous build
        self.uses = 0
        self.flat: Optional[RowGather] = None
        self.own = {}   # derived index arrays of the pattern itself (COO → crow, stable row order, ...)

    def nbytes(self) -> int:
        seen, total = set(), 0

        def add(t):
            nonlocal total
            if torch.is_tensor(t) and t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                total += t.numel() * t.element_size()

        add(self.rows)
        for t in self.own.values():
            if isinstance(t, tuple):
                for u in t:
                    add(u)  # (the core's copy of its index tensors)
            else:
                add(t)  # (non-tensor entries are ignored)
        for rp in self.packs.values():
            if rp is not None:
                total += rp.plan_bytes()
        for sub in (self.t, self.flat):
            if sub is not None:
                add(sub.crow), add(sub.col), add(sub.perm)
                total += sub.core.nbytes()
        return total


class RowGather:
    """Row-gather structure of a (batched) sparse matrix on the device.

    crow: (n_rows+1,) or (b, n_rows+1); col: (nnz,) or (b, nnz); perm: optional, same shape as
    col, position of each entry in the owner's value array (None = identity).  Derived analyses
    live in ``core`` (shared between views of the same pattern through the module cache).
    """

    __slots__ = ("crow", "col", "perm", "n_rows", "n_cols", "batch", "core")

    def __init__(self, crow, col, n_rows, n_cols, perm=None, core: Optional[_Core] = None):
        self.crow, self.col, self.perm = crow, col, perm
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.batch = crow.size(0) if crow.dim() == 2 else None
        self.core = _Core() if core is None else core

    # kept as attributes of the view for the tests / tools that look at them
    @property
    def _packs(self):
        return self.core.packs

    def seen_enough(self, after: int) -> bool:
        """Count one use of this pattern; True once it has been used more than `after` times (plan policy:
        expensive plans are only built for patterns that come back)."""
        self.core.uses += 1
        return self.core.uses > after

    def rowpack_plan(self, rows_per_block: int, limits, explicit_slots: bool = False, group: int = 2):
        """Plan for the row-pair (group = 2) / row-quad (group = 4) gather kernels (csrc/rowpack_impl.h), None when the
        pattern does not qualify or profit; cached per workgroup height.  See `build_rowpack_plan`."""
        key = (rows_per_block, tuple(limits), bool(explicit_slots) or self.perm is not None, group)
        packs = self.core.packs
        if key not in packs:
            fut = self.core.pending.get(key)
            if fut is not None:           # an asynchronous build is in flight: wait for it rather than build twice
                _wait_quietly(fut)        # (a failed build is reported — as a warning — by _plan_async, which pins the pattern plan-free)
                return self.rowpack_plan_async(rows_per_block, limits, explicit_slots, group)
            packs[key] = self._build_rowpack(rows_per_block, limits, explicit_slots, group)
        return packs[key]

    def rowpack_plan_async(self, rows_per_block: int, limits, explicit_slots: bool = False, group: int = 2):
        """Like `rowpack_plan`, but the plan is built on a worker thread + side stream: returns the plan once it is
        ready (None until then, so that the caller keeps running the plan-free kernels instead of stalling a training
        step for the plan's device sorts).  Results do not depend on when the switch happens beyond the documented
        difference between the two kernel families."""
        key = (rows_per_block, tuple(limits), bool(explicit_slots) or self.perm is not None, group)
        return self._plan_async(key, lambda view: view._build_rowpack(rows_per_block, limits, explicit_slots, group),
                                lambda plan: (plan.uptr, plan.ucol, plan.upos, plan.sperm, plan.order, plan.vpair, plan.eptr, plan.wcls,
                                              plan.wbase, plan.cne, plan.srcstart))

    def tile_plan(self, geo, asynchronous: bool = False):
        """Plan of the row-block tile kernels (csrc/tile_impl.h; `geo` = (rows per block, max distinct columns, max entries) from
        tsgu_tile_geometry), None when the pattern does not qualify; cached with the pattern.  `asynchronous`: built on the worker
        thread + side stream, None until it is ready (see rowpack_plan_async)."""
        key = ("tile",) + tuple(geo)
        if asynchronous:
            return self._plan_async(key, lambda view: view._build_tile(geo), lambda plan: (plan.desc, plan.ucol, plan.lidx, plan.rptr, plan.cpos, plan.cslot, plan.ent, plan.xrow))
        packs = self.core.packs
        if key not in packs:
            fut = self.core.pending.get(key)
            if fut is not None:
                _wait_quietly(fut)        # (see rowpack_plan: the worker's exception must not reach the caller's step)
                return self.tile_plan(geo, asynchronous=True)
            packs[key] = self._build_tile(geo)
        return packs[key]

    def _build_tile(self, geo):
        from . import _tile

        if self.batch is not None or self.crow.dtype not in (torch.int32, torch.int64):
            return None
        return _tile.build_tile_plan(self.crow, self.col, self.n_rows, self.n_cols, geo[0], geo[1], geo[2], perm=self.perm,
                                     rows=self.row_indices())

    def _plan_async(self, key, build, tensors_of):
        """Generic asynchronous plan build: `build(view)` runs on the worker thread's side stream, the plan is handed out (and its
        tensors `tensors_of(plan)` are handed to the caller's stream) by the first call after it has finished."""
        core = self.core
        if key in core.packs:
            return core.packs[key]
        dev = self.crow.device
        fut = core.pending.get(key)
        if fut is None:
            if torch.cuda.is_current_stream_capturing():
                return None      # no cross-stream event inside a graph capture: the plan is asked for again after it
            # everything the builder reads lazily from the shared core is produced HERE, on the caller's stream, before the
            # event: the worker's side stream then only reads it (no unsynchronised write of core.rows from two streams)
            self.row_indices()
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))   # the index tensors are complete from here on
            view = RowGather(self.crow, self.col, self.n_rows, self.n_cols, perm=self.perm, core=core)

            def job():
                with torch.cuda.device(dev):
                    side = _side_stream(dev)
                    with torch.cuda.stream(side):
                        side.wait_event(ready)
                        plan = build(view)
                        done = torch.cuda.Event()
                        done.record(side)
                return plan, done

            with _PENDING_LOCK:
                core.pending[key] = _EXECUTOR.submit(job)
                _PENDING_CORES.add(core)
            return None
        if not fut.done():
            return None
        try:
            plan, done = fut.result()
        except Exception as exc:  # noqa: BLE001  (a failed background build must never break the caller's step)
            warnings.warn(f"torchsparsegradutils_amd: asynchronous plan build failed ({exc!r}); "
                          "this pattern stays on the plan-free kernels", RuntimeWarning, stacklevel=2)
            with _PENDING_LOCK:
                core.packs[key] = None
                core.pending.pop(key, None)
                if not core.pending:
                    _PENDING_CORES.discard(core)
            return None
        if torch.cuda.is_current_stream_capturing():
            return None          # picked up by the first call outside the capture
        main = torch.cuda.current_stream(dev)
        main.wait_event(done)
        if plan is not None:
            for t in tensors_of(plan):
                if t is not None:
                    t.record_stream(main)   # allocated on the side stream, used on the caller's from now on
        with _PENDING_LOCK:
            core.packs[key] = plan
            core.pending.pop(key, None)
            if not core.pending:
                _PENDING_CORES.discard(core)
        return plan

    def _build_rowpack(self, rows_per_block: int, limits, explicit_slots: bool, group: int = 2):
        if group != 2:
            return build_rowpack_plan(self, rows_per_block, limits, group=group)
        plan = None
        if ENABLE_BRICKS and self.perm is not None:
            lat = detect_lattice(self)
            order = brick_pair_order(self.n_rows, lat, rows_per_block // 2, self.crow.device) if lat else None
            if order is not None:
                plan = build_rowpack_plan(self, rows_per_block, limits, pair_order=order, lattice=lat)
        if plan is None:
            plan = build_rowpack_plan(self, rows_per_block, limits, explicit_slots=explicit_slots)
        return plan

    @property
    def nnz(self) -> int:
        return self.col.size(-1)

    def row_indices(self) -> torch.Tensor:
        """Expanded row index per stored entry (same shape/dtype as col); cached."""
        if self.core.rows is None:
            n = self.n_rows
            ar = torch.arange(n, dtype=self.col.dtype, device=self.col.device)
            if self.batch is None:
                self.core.rows = torch.repeat_interleave(ar, self.crow[1:] - self.crow[:-1], output_size=self.nnz)
            else:
                counts = (self.crow[:, 1:] - self.crow[:, :-1]).reshape(-1)
                rows = torch.repeat_interleave(ar.repeat(self.batch), counts, output_size=self.batch * self.nnz)
                self.core.rows = rows.view(self.batch, self.nnz)
        return self.core.rows

    @property
    def max_row_nnz(self) -> int:
        """Length of the longest row (one small reduction + host read per pattern, cached)."""
        m = self.core.own.get("max_row_nnz")
        if m is None:
            d = self.crow[..., 1:] - self.crow[..., :-1]
            m = self.core.own["max_row_nnz"] = int(d.max()) if d.numel() else 0
        return m

    @property
    def transposed(self) -> "RowGather":
        """Row-gather structure of Aᵀ whose ``perm`` indexes A's value array."""
        if self.core.t is None:
            self.core.t = _transpose(self)
        return self.core.t

    @property
    def has_diagonal(self) -> bool:
        if self.core.has_diag is None:
            self.core.has_diag = bool(torch.any(self.row_indices() == self.col))
        return self.core.has_diag


class RowPackPlan:
    """Union-of-columns walk for pairs of consecutive rows, consumed by tsgu_csr_*_rowpack (layout: include/tsgu_hip.h).

    Two storage forms of the same walk:
    * stream form — ``uptr / ucol / upos / sperm`` hold one record per union entry / stored entry of the whole matrix;
    * class-dictionary form (``nclasses > 0``) — workgroups whose records are translations of each other (equal after
      subtracting the workgroup's base pair / base column / base value position) share ONE copy of the records:
      ``uptr / ucol / upos / sperm / vpair`` are then per-class tables with fixed strides and ``wcls`` / ``wbase`` give
      every workgroup its class and its three bases.  On lattice stencils (a few dozen classes for millions of rows)
      the index streams — a third of the kernels' HBM traffic — become L2-resident."""

    __slots__ = ("uptr", "ucol", "upos", "sperm", "order", "vpair", "eptr", "nblocks", "ecap", "ucap", "rpb", "reuse",
                 "nnz", "lattice", "wcls", "wbase", "cne", "nclasses", "gpb", "group", "srcstart", "_cstruct")

    def __init__(self, uptr, ucol, upos, sperm, ecap, ucap, rpb, reuse, nnz, order=None, vpair=None, eptr=None,
                 nblocks=0, lattice=None, gpb=None):
        self.uptr, self.ucol, self.upos, self.sperm, self.order = uptr, ucol, upos, sperm, order
        self.vpair, self.eptr, self.nblocks, self.lattice = vpair, eptr, nblocks, lattice
        self.ecap, self.ucap, self.rpb, self.reuse, self.nnz = ecap, ucap, rpb, reuse, nnz
        self.gpb = rpb // 2 if gpb is None else gpb
        self.group = 2          # rows per lane group (2 = pairs, 4 = quads)
        self.wcls = self.wbase = self.cne = None
        self.nclasses = 0
        self.srcstart = None     # dictionary form of a permuted plan: first value position of every source row (see _dedup_classes)
        self._cstruct = None

    def plan_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.uptr, self.ucol, self.upos, self.sperm, self.order, self.vpair,
                                                          self.eptr, self.wcls, self.wbase, self.cne, self.srcstart) if t is not None)


_PACK_MIN_REUSE = 1.2   # stored entries per union entry (2.0 = both rows of every pair share all columns)
_PACK_ABSENT = 0x8000
ENABLE_BRICKS = True    # lattice patterns: permuted walks own 3-D bricks of row pairs (see brick_pair_order)
BRICK_ROTATE = _os.environ.get("TSGU_BRICK_ROTATE", "1") == "1"   # plane-rotated record order inside bricks (L1 reuse between waves)


def detect_lattice(g: RowGather):
    """Strides (d1,) or (d1, d2) of a row-major 2-D / 3-D lattice whose stencil this square pattern is (row =
    x·d2 + y·d1 + z), or None.  Heuristic on the histogram of col − row: the offsets present in at least half of
    the rows form clusters {±1}, {d1 − r … d1 + r}, {d2 − r' … d2 + r'}; the cluster centres are the strides.
    Wrap-around offsets of periodic stencils are rare and ignored.  Anything irregular returns None."""
    if g.batch is not None or g.n_rows != g.n_cols or g.n_rows < 64 or g.nnz == 0:
        return None
    n = g.n_rows
    off = g.col.to(torch.int64) - g.row_indices().to(torch.int64)
    uniq, cnt = torch.unique(off, return_counts=True)
    if uniq.numel() > 4096:
        return None
    pos = uniq[(cnt >= n // 2) & (uniq > 0)].tolist()  # few values: host side
    if not pos or pos[0] != 1:
        return None
    clusters, reach = [[pos[0]]], 1
    for o in pos[1:]:
        if o - clusters[-1][-1] <= reach:
            clusters[-1].append(o)
        else:
            reach = clusters[-1][-1]
            clusters.append([o])
    if len(clusters) not in (2, 3) or clusters[0][-1] != 1:
        return None
    strides = []
    for c in clusters[1:]:
        centre2 = c[0] + c[-1]
        if centre2 % 2:
            return None
        strides.append(centre2 // 2)
    d1 = strides[0]
    if d1 < 4 or d1 % 2 or n % d1:
        return None
    if len(strides) == 2:
        d2 = strides[1]
        if d2 % d1 or n % d2 or d2 // d1 < 2 or n // d2 < 2:
            return None
        return (d1, d2)
    return (d1,) if n // d1 >= 2 else None


def brick_pair_order(n: int, lattice, gpb: int, device, shape=None):
    """Lane-group slot → row pair so that a workgroup of `gpb` slots owns a brick of the lattice (pairs run along z).
    A brick's entries form long runs in the transposed operand's value array (all 27 neighbours of an interior
    point belong to it), instead of the 3-word runs a block of consecutive rows gives.  Returns an int64 tensor of
    nblocks·gpb pair indices, -1 where a brick sticks out of the lattice."""
    d1 = lattice[0]
    nzp = d1 // 2
    if len(lattice) == 2:
        d2 = lattice[1]
        ny, nx = d2 // d1, n // d2
        if shape is None:
            shape = {16: (4, 2, 2), 32: (4, 2, 4), 64: (4, 4, 4)}.get(gpb)  # (z pairs, y, x); C2: 312 -> 250-263 us
    else:
        d2, ny, nx = n, n // d1, 1
        if shape is None:
            shape = {16: (4, 4, 1), 32: (8, 4, 1), 64: (8, 8, 1)}.get(gpb)
    if shape is None or shape[0] * shape[1] * shape[2] != gpb:
        return None
    pz, by, bx = shape
    ar = lambda k: torch.arange(k, device=device, dtype=torch.int64)  # noqa: E731
    Xb, Yb, Zb = -(-nx // bx), -(-ny // by), -(-nzp // pz)
    X, Y, Z, ix, iy, ip = torch.meshgrid(ar(Xb), ar(Yb), ar(Zb), ar(bx), ar(by), ar(pz), indexing="ij")
    x, y, zp = X * bx + ix, Y * by + iy, Z * pz + ip
    ok = (x < nx) & (y < ny) & (zp < nzp)
    pair = (x * d2 + y * d1) // 2 + zp
    return torch.where(ok, pair, torch.full_like(pair, -1)).reshape(-1)


DEDUP_MODE = _os.environ.get("TSGU_DEDUP", "auto")   # "auto" | "off" | "force" (tests: small matrices have few workgroups)
DEDUP_MAX_FRACTION = 0.25    # dictionary form when the classes are at most this fraction of the workgroups ...
DEDUP_MAX_BYTES = 8 << 20    # ... and the class tables stay cache-sized
ROW_RELATIVE = _os.environ.get("TSGU_DEDUP_ROW_RELATIVE", "1") == "1"   # permuted plans: value positions relative to the source row


def build_rowpack_plan(g: RowGather, rpb: int, limits, pair_order=None, lattice=None, explicit_slots=False, dedup=None,
                       group: int = 2):
    """Rows 2q and 2q+1 walk the sorted union of their column sets: `ucol` per union entry, `upos` = two 16-bit slots
    (one per row; bit 15 = this row has no entry there) into the value slice a workgroup of `rpb` rows stages.  For
    plans addressed through `perm` the slice is staged in the order of the permutation sorted inside the workgroup
    (`sperm`), otherwise in stored order — and then, unless `explicit_slots` (kernels with several entry lanes per
    pair), the record only carries ownership bits (top bits of ucol) because a row's slots are consecutive.
    `group` = 4 builds the ROW-QUAD plan (stored order, no slots): a lane group owns four consecutive rows.
    `pair_order` (permuted plans only) assigns row pairs to lane-group slots (see brick_pair_order); None =
    consecutive.  `limits` = (max_entries, max_union, lds_budget_bytes).  `dedup` overrides DEDUP_MODE."""
    max_entries, max_union, lds_budget = limits
    if g.batch is not None or g.n_rows == 0 or not (1 <= g.nnz < 2**31):
        return None
    n, m, nnz = g.n_rows, g.n_cols, g.nnz
    dev = g.crow.device
    slots = explicit_slots or g.perm is not None
    if group != 2:
        return None      # (row quads, group = 4, are no longer compiled: csrc/rowpack_impl.h)
    gpb = rpb // group
    npairs = (n + group - 1) // group             # row groups (pairs / quads)
    natural = pair_order is None
    if natural:
        nb = (npairs + gpb - 1) // gpb
        pair_order = torch.full((nb * gpb,), -1, dtype=torch.int64, device=dev)
        pair_order[:npairs] = torch.arange(npairs, device=dev)
    elif g.perm is None or pair_order.numel() % gpb:
        return None
    nslots = pair_order.numel()
    nb = nslots // gpb
    valid = pair_order >= 0
    slot_of = torch.full((npairs,), -1, dtype=torch.int64, device=dev)
    slot_of[pair_order[valid]] = torch.nonzero(valid).flatten()
    if int(valid.sum()) != npairs or bool((slot_of < 0).any()):
        return None  # not a permutation of the pairs
    rows = g.row_indices().to(torch.int64)
    if nnz > 1:
        # the union walk visits a row's entries in ascending column order: stored rows must be sorted and duplicate-free
        c64 = g.col.to(torch.int64)
        if bool(((c64[1:] <= c64[:-1]) & (rows[1:] == rows[:-1])).any()):
            return None
    vp = slot_of[rows // group]                   # lane-group slot of every stored entry
    blk = vp // gpb
    ne = torch.bincount(blk, minlength=nb)
    ecap = max((int(ne.max()) + 255) // 256 * 256, 256)
    if ecap > max_entries * (group // 2) or ecap >= _PACK_ABSENT:
        return None
    eptr = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    eptr[1:] = torch.cumsum(ne, 0)
    uniq, inv = torch.unique(vp * m + g.col.to(torch.int64), return_inverse=True)  # sorted: (slot, column) ascending
    nu = uniq.numel()
    reuse = nnz / max(nu, 1)
    if reuse < _PACK_MIN_REUSE * (1.0 if group == 2 else 1.25):
        return None
    uslot = uniq // m
    if BRICK_ROTATE and slots and not natural and lattice is not None and len(lattice) == 2:
        # Brick plans of a 3-D lattice: the four waves of a workgroup own the brick's four x-planes and each needs the
        # planes x-1, x, x+1.  In ascending column order wave w reaches plane p a third of the walk after wave w+1 did —
        # long enough for L1 to have lost it.  Rotating every pair's records by planes (wave w visits its relative
        # plane rp in slot (rp + 1 + w) mod 3) makes the waves that share a plane fetch it at the same time; records with
        # explicit slots may come in any order (only the order of summation inside a row changes).
        d2 = lattice[1]
        nxl = m // d2
        ucolr = uniq - uslot * m
        x_own = (2 * pair_order[uslot]) // d2
        rp_ = (ucolr // d2 - x_own + nxl + 1) % nxl - 1          # relative plane in {-1, 0, 1} (periodic wrap)
        wv = (uslot % gpb) // max(gpb // 4, 1)
        phase = torch.where((rp_ >= -1) & (rp_ <= 1), (rp_ + 1 + wv) % 3, torch.full_like(rp_, 3))
        ordr = torch.argsort((uslot * 4 + phase) * m + ucolr)
        rank = torch.empty_like(ordr)
        rank[ordr] = torch.arange(ordr.numel(), device=dev)
        uniq = uniq[ordr]
        inv = rank[inv]
        uslot = uniq // m
    uptr = torch.zeros(nslots + 1, dtype=torch.int64, device=dev)
    uptr[1:] = torch.cumsum(torch.bincount(uslot, minlength=nslots), 0)
    ub = uptr[torch.arange(0, nslots + 1, gpb, device=dev)]
    ucap = max((int((ub[1:] - ub[:-1]).max()) + 255) // 256 * 256, 256)
    if ucap > max_union or ucap * (8 if slots else 4) + ecap * 4 > lds_budget or (not slots and m >= 2 ** (32 - group)):
        return None
    k = torch.arange(nnz, device=dev, dtype=torch.int64)
    sperm64 = srow64 = None
    if g.perm is None:
        slot = k - eptr[blk]                      # natural order: a workgroup's entries are one contiguous range
    else:
        order = torch.argsort(blk * nnz + g.perm.to(torch.int64))
        sperm64 = g.perm[order].to(torch.int64)
        srow64 = g.col[order].to(torch.int64)      # the source row of every staged value (a permuted plan walks the transposed pattern)
        slot = torch.empty(nnz, dtype=torch.int64, device=dev)
        slot[order] = k - eptr[blk[order]]
    ucol64 = uniq - uslot * m
    own = word64 = None
    if not slots:
        # stored order: a row's slots are consecutive, the record only carries the ownership bits (top `group` bits)
        bits = torch.zeros((group, nu), dtype=torch.int64, device=dev)
        bits[rows % group, inv] = 1
        own = torch.zeros(nu, dtype=torch.int64, device=dev)
        for r in range(group):
            own |= bits[r] << (32 - group + r)
    else:
        half = torch.full((2, nu), _PACK_ABSENT, dtype=torch.int64, device=dev)
        half[rows % 2, inv] = slot
        word64 = half[0] | (half[1] << 16)
    plan = RowPackPlan(None, None, None, None, ecap, ucap, rpb, reuse, nnz, nblocks=nb, lattice=lattice, gpb=gpb)
    plan.group = group
    mode = DEDUP_MODE if dedup is None else dedup
    if mode != "off" and _dedup_classes(plan, mode == "force", natural, uptr, ub, ucol64, own, word64, sperm64, pair_order,
                                        eptr, nb, gpb, srow64, m):
        return plan
    plan.uptr = uptr.to(torch.int32).contiguous()
    plan.ucol = _to_i32(ucol64 | own if own is not None else ucol64)
    plan.upos = None if word64 is None else _to_i32(word64)
    plan.sperm = None if sperm64 is None else sperm64.to(torch.int32).contiguous()
    if not natural:
        plan.vpair = pair_order.to(torch.int32).contiguous()
        plan.eptr = eptr.to(torch.int32).contiguous()
    return plan


def _to_i32(w: torch.Tensor) -> torch.Tensor:
    """int64 holding a uint32 bit pattern -> int32 tensor with the same bits."""
    return torch.where(w >= 2**31, w - 2**32, w).to(torch.int32).contiguous()


def _mix(a: torch.Tensor, b, pos) -> torch.Tensor:
    """64-bit mixing of (a, b, position) with wrap-around integer arithmetic (collisions are caught by the exact
    comparison in _dedup_classes, so this only has to be well spread)."""
    if not torch.is_tensor(pos):
        pos = torch.tensor(pos, dtype=torch.int64, device=a.device)
    x = a * -7046029254386353131 + b * -4417276706812531889 + 1609587929392839161
    x = x ^ (pos * 2870177450012600261 + 7046029254386353131)
    return x * -49064778989728563


def _dedup_classes(plan: RowPackPlan, force: bool, natural: bool, uptr, ub, ucol64, own, word64, sperm64, pair_order, eptr,
                   nb: int, gpb: int, srow64=None, n_src: int = 0) -> bool:
    """Translation-deduplicate the per-workgroup records of a row-pair plan (see RowPackPlan).  Fills the
    class-dictionary fields of `plan` and returns True when the dictionary form is used."""
    dev = uptr.device
    i64 = torch.int64
    big = torch.iinfo(i64).max
    ar_nb = torch.arange(nb, device=dev)
    nu_b = ub[1:] - ub[:-1]
    nu = int(ub[-1])
    ne_b = eptr[1:] - eptr[:-1]
    nnz = int(eptr[-1])
    wg_u = torch.repeat_interleave(ar_nb, nu_b, output_size=nu)
    pos_u = torch.arange(nu, device=dev) - ub[wg_u]
    base_col = torch.full((nb,), big, dtype=i64, device=dev).scatter_reduce_(0, wg_u, ucol64, "amin")
    base_col = torch.where(nu_b > 0, base_col, torch.zeros_like(base_col))
    rel_col = ucol64 - base_col[wg_u]
    po = pair_order.view(nb, gpb)
    valid = po >= 0
    base_pair = torch.where(valid, po, torch.full_like(po, big)).amin(1)
    base_pair = torch.where(valid.any(1), base_pair, torch.zeros_like(base_pair))
    rel_pair = torch.where(valid, po - base_pair[:, None], torch.full_like(po, -1))
    rel_uptr = uptr[:-1].view(nb, gpb) - ub[:-1, None]
    rec = word64 if word64 is not None else own
    h = torch.zeros(nb, dtype=i64, device=dev)
    h.index_add_(0, wg_u, _mix(rel_col, rec, pos_u))
    h += _mix(rel_pair, rel_uptr, torch.arange(gpb, device=dev)[None, :] + 1000003).sum(1)
    h += _mix(nu_b, ne_b, 7)
    rel_perm = base_perm = None
    if sperm64 is not None:
        wg_e = torch.repeat_interleave(ar_nb, ne_b, output_size=nnz)
        pos_e = torch.arange(nnz, device=dev) - eptr[wg_e]
        base_perm = torch.zeros(nb, dtype=i64, device=dev)
        has = ne_b > 0
        srcstart = None
        if srow64 is not None and ROW_RELATIVE:
            # Positions relative to the workgroup's first value differ between translated workgroups wherever rows of another length
            # lie in between (a mesh with truncated rows at its faces: 445 classes for 1000 workgroups).  Relative to the START OF ITS
            # SOURCE ROW a value's position is translation-invariant: the record is (source row - the workgroup's first source row,
            # offset inside the row) and the kernel adds the row's first position, which the plan keeps per source row.
            srcstart = torch.full((n_src,), nnz, dtype=i64, device=dev).scatter_reduce_(0, srow64, sperm64, "amin")
            off = sperm64 - srcstart[srow64]
            base_perm[has] = srow64[eptr[:-1][has]]      # (positions ascend inside a workgroup, and so do their rows)
            rel_row = srow64 - base_perm[wg_e]
            if int(off.max()) < 256 and int(rel_row.min()) >= 0 and int(rel_row.max()) < (1 << 23):
                rel_perm = (rel_row << 8) | off
            else:
                srcstart = None
        if srcstart is None:
            base_perm[has] = sperm64[eptr[:-1][has]]   # sorted ascending inside a workgroup: the first is the smallest
            rel_perm = sperm64 - base_perm[wg_e]
        h.index_add_(0, wg_e, _mix(rel_perm, pos_e, 3))
    uniq, inv = torch.unique(h, return_inverse=True)
    ncls = uniq.numel()
    slots = word64 is not None
    table_bytes = ncls * ((gpb + 1) * 4 + plan.ucap * (8 if slots else 4) + (plan.ecap * 4 if sperm64 is not None else 0) + gpb * 4)
    if not force and (ncls > DEDUP_MAX_FRACTION * nb or table_bytes > DEDUP_MAX_BYTES):
        return False
    rep = torch.full((ncls,), nb, dtype=i64, device=dev).scatter_reduce_(0, inv, ar_nb, "amin")
    r_b = rep[inv]
    # exact check against the class representative (a hash collision falls back to the stream form)
    ok = (torch.equal(nu_b, nu_b[r_b]) and torch.equal(ne_b, ne_b[r_b]) and torch.equal(rel_pair, rel_pair[r_b])
          and torch.equal(rel_uptr, rel_uptr[r_b]))
    if ok:
        idx = ub[r_b][wg_u] + pos_u
        ok = torch.equal(rel_col, rel_col[idx]) and torch.equal(rec, rec[idx])
    if ok and sperm64 is not None:
        idx_e = eptr[r_b][wg_e] + pos_e
        ok = torch.equal(rel_perm, rel_perm[idx_e])
    if not ok:
        return False
    cuptr = torch.zeros((ncls, gpb + 1), dtype=i64, device=dev)
    cuptr[:, :gpb] = rel_uptr[rep]
    cuptr[:, gpb] = nu_b[rep]
    is_rep = r_b == ar_nb
    mu = is_rep[wg_u]
    cu, pu = inv[wg_u[mu]], pos_u[mu]
    cucol = torch.zeros((ncls, plan.ucap), dtype=i64, device=dev)
    cucol[cu, pu] = rel_col[mu] | own[mu] if own is not None else rel_col[mu]
    plan.uptr = cuptr.to(torch.int32).reshape(-1).contiguous()
    plan.ucol = _to_i32(cucol.reshape(-1))
    if slots:
        cupos = torch.full((ncls, plan.ucap), _PACK_ABSENT | (_PACK_ABSENT << 16), dtype=i64, device=dev)
        cupos[cu, pu] = word64[mu]
        plan.upos = _to_i32(cupos.reshape(-1))
    if sperm64 is not None:
        me = is_rep[wg_e]
        csperm = torch.zeros((ncls, plan.ecap), dtype=i64, device=dev)
        csperm[inv[wg_e[me]], pos_e[me]] = rel_perm[me]
        plan.sperm = csperm.to(torch.int32).reshape(-1).contiguous()
        plan.cne = ne_b[rep].to(torch.int32).contiguous()
        plan.srcstart = None if srcstart is None else srcstart.clamp_(max=nnz).to(torch.int32).contiguous()
    if not natural:
        plan.vpair = rel_pair[rep].to(torch.int32).reshape(-1).contiguous()
    wbase = torch.stack((base_pair, base_col, base_perm if base_perm is not None else torch.zeros_like(base_pair)), 1)
    plan.wbase = wbase.to(torch.int32).contiguous()
    plan.wcls = inv.to(torch.int32).contiguous()
    plan.nclasses = ncls
    return True


def expand_classes(plan: RowPackPlan):
    """Stream-form arrays (uptr, ucol, upos, sperm, vpair, eptr) of a class-dictionary plan — what the kernels see
    after adding every workgroup's bases.  Test / debugging helper."""
    assert plan.nclasses > 0
    gpb, nb = plan.gpb, plan.nblocks
    cls = plan.wcls.long()
    wb = plan.wbase.long()
    cuptr = plan.uptr.view(-1, gpb + 1).long()[cls]                     # (nb, gpb+1) relative
    nu_b = cuptr[:, gpb]
    ub = torch.zeros(nb + 1, dtype=torch.int64, device=cls.device)
    ub[1:] = torch.cumsum(nu_b, 0)
    uptr = torch.cat(((cuptr[:, :gpb] + ub[:-1, None]).reshape(-1), ub[-1:]))
    ar = torch.arange(plan.ucap, device=cls.device)[None, :]
    mu = ar < nu_b[:, None]
    slots = plan.upos is not None
    words = plan.ucol.view(-1, plan.ucap).long()[cls] & 0xFFFFFFFF
    if slots:
        ucol = (words + wb[:, 1:2])[mu]
        upos = (plan.upos.view(-1, plan.ucap).long()[cls] & 0xFFFFFFFF)[mu]
    else:
        cmask = (1 << (32 - plan.group)) - 1
        ucol = (((words & cmask) + wb[:, 1:2]) | (words & (0xFFFFFFFF ^ cmask)))[mu]
        upos = None
    sperm = eptr = None
    if plan.sperm is not None:
        ne_b = plan.cne.long()[cls]
        eptr = torch.zeros(nb + 1, dtype=torch.int64, device=cls.device)
        eptr[1:] = torch.cumsum(ne_b, 0)
        me = torch.arange(plan.ecap, device=cls.device)[None, :] < ne_b[:, None]
        code = plan.sperm.view(-1, plan.ecap).long()[cls]
        if plan.srcstart is not None:      # (source row relative to the workgroup's first, offset inside the row)
            sperm = (plan.srcstart.long()[((code >> 8) + wb[:, 2:3]).clamp_(max=plan.srcstart.numel() - 1)] + (code & 0xFF))[me]
        else:
            sperm = (code + wb[:, 2:3])[me]
    if plan.vpair is not None:
        rel = plan.vpair.view(-1, gpb).long()[cls]
        vpair = torch.where(rel >= 0, rel + wb[:, 0:1], rel).reshape(-1)
    else:
        vpair = (wb[:, 0:1] + torch.arange(gpb, device=cls.device)[None, :]).reshape(-1)
    return uptr, ucol, upos, sperm, vpair, eptr


def _transpose(g: RowGather) -> RowGather:
    rows = g.row_indices()
    idt = g.col.dtype
    if g.batch is None:
        order = torch.argsort(g.col, stable=True)
        counts = torch.bincount(g.col, minlength=g.n_cols)
        ptr = torch.zeros(g.n_cols + 1, dtype=idt, device=g.col.device)
        ptr[1:] = torch.cumsum(counts, 0)
        idx = rows[order]
        perm = order if g.perm is None else g.perm[order]
        return RowGather(ptr, idx.contiguous(), g.n_cols, g.n_rows, perm.to(idt).contiguous())
    b, nnz = g.batch, g.nnz
    item = torch.arange(b, device=g.col.device, dtype=torch.int64).unsqueeze(1)
    key = (g.col.to(torch.int64) + item * g.n_cols).reshape(-1)
    order = torch.argsort(key, stable=True)
    counts = torch.bincount(key, minlength=b * g.n_cols).view(b, g.n_cols)
    ptr = torch.zeros((b, g.n_cols + 1), dtype=idt, device=g.col.device)
    ptr[:, 1:] = torch.cumsum(counts, 1)
    idx = rows.reshape(-1)[order].view(b, nnz)
    local = (order.view(b, nnz) - item * nnz)
    perm = local if g.perm is None else torch.gather(g.perm.to(torch.int64), 1, local)
    return RowGather(ptr, idx.contiguous(), g.n_cols, g.n_rows, perm.to(idt).contiguous())


# ---- cache ---------------------------------------------------------------------------------
# key -> _Core.  The key identifies the caller's index tensors by storage identity + view geometry + version; the
# entry is evicted when any of those storages dies (weakref finalizer: torch keeps one Python object per live
# storage), when the LRU count is exceeded, or when the derived bytes exceed the budget.  Cores never reference
# the caller's tensors, so `del A` really frees the plans.

_CACHE: "OrderedDict[tuple, _Core]" = OrderedDict()
_CACHE_LOCK = threading.RLock()  # backward runs on autograd threads; finalizers may run inside a locked region
_CACHE_MAX = 16
# patterns whose index tensors have all died, kept adoptable (see _evict); TSGU_PLAN_CACHE_RECENT=0: plans die with their tensors
_RECENT: "OrderedDict[int, _Core]" = OrderedDict()
RECENT_MAX = int(_os.environ.get("TSGU_PLAN_CACHE_RECENT", "2"))
_CACHE_MAX_BYTES = int(_os.environ.get("TSGU_PLAN_CACHE_BYTES", str(8 << 30)))


def _key(kind: str, tensors, shape) -> tuple:
    # (torch.Size, the stride tuple and torch.device hash as they are: this runs once per product on the host's critical path)
    return (kind, shape if isinstance(shape, tuple) else tuple(shape)) + tuple(
        (t.untyped_storage()._cdata, t.storage_offset(), t.shape, t.stride(), t.dtype, t._version, t.device)
        for t in tensors)


def _evict(key) -> None:
    with _CACHE_LOCK:
        core = _CACHE.pop(key, None)
        # A caller that rebuilds its index tensors every step usually DROPS the previous ones before the next call: the entry of the
        # old tensors dies here, one step before tensors with the same content arrive.  The last few patterns that lost their last key
        # stay adoptable for a while (strong references, bounded in number and by the cache's byte budget): without this the
        # recognition of fresh tensors only ever worked for callers that kept the old tensors alive (bench.py's and the tests' loops did).
        if core is not None and "index_copy" in core.own and not any(c is core for c in _CACHE.values()):
            _RECENT[id(core)] = core
            _RECENT.move_to_end(id(core))
            while len(_RECENT) > RECENT_MAX:
                _RECENT.popitem(last=False)


# A caller that rebuilds its index tensors every step (`torch.sparse_csr_tensor(crow.clone(), col.clone(), …)`) misses the identity
# key every time: without more, every step would pay the pattern analysis (~11 ms at C2) instead of the 0.7 ms plan-free step.
# The reference has no per-pattern state and so no such cliff (sparse_matmul.py:141-163).  On a miss whose geometry (layout, shape,
# index dtype, nnz, device) equals a live entry's, the CONTENT of the index tensors is compared with that entry's — EXACTLY: every
# core keeps its own copy of the index tensors it was built from (`own["index_copy"]`, written by the pass that fingerprints them),
# and one pass over the new tensors (`tsgu_index_fingerprint_match`) yields their 128-bit fingerprint AND whether they equal the most
# recently used candidate's copy; the answer is one host read.  Equal → the live core is adopted under the new key.  Otherwise the
# fingerprint selects among the other candidates, and the selected one is compared exactly before it is adopted: equal fingerprints
# alone never adopt (a collision would silently compute with another matrix' pattern).  Geometries that keep arriving with NEW content
# are marked volatile after FRESH_LIMIT misses in a row: their plans then wait until a pattern has come back
# (`_Core.own["volatile"]`, read by _ops._lattice_plan).  What a core's fingerprint words and copies were written by is recorded
# as an event: a miss on another stream waits for it before it reads them.
FINGERPRINT = _os.environ.get("TSGU_PATTERN_FINGERPRINT", "1") != "0"
FINGERPRINT_MIN = 1 << 16
# index tensors above this many bytes are not copied (and their patterns never adopted): the copy lives as long as the cache entry
INDEX_COPY_MAX_BYTES = int(_os.environ.get("TSGU_INDEX_COPY_MAX_BYTES", str(2 << 30)))
FRESH_LIMIT = 3
_FRESH = {}
STATS = {"adopted": 0, "fingerprints": 0, "volatile": 0, "verified": 0, "collisions": 0}


_READ_BUFS = threading.local()


def _read_words_begin(t: torch.Tensor):
    """Queue the copy of a small int64 device tensor into pinned memory (behind everything the current stream holds NOW: what is
    queued after this call does not delay the answer); _read_words_end waits for it."""
    n = t.numel()
    bufs = getattr(_READ_BUFS, "bufs", None)
    if bufs is None:
        bufs = _READ_BUFS.bufs = {}
    dev = t.device
    got = bufs.get(dev)
    if got is None or got[0].numel() < n:
        got = bufs[dev] = (torch.empty(max(n, 64), dtype=torch.int64, pin_memory=True), torch.cuda.Event())
    host, ev = got
    host[:n].copy_(t.reshape(-1), non_blocking=True)
    ev.record(torch.cuda.current_stream(dev))
    return host, ev, n


def _read_words_end(handle) -> list:
    host, ev, n = handle
    while not ev.query():
        pass
    return host[:n].tolist()


def _read_words(t: torch.Tensor) -> list:
    """The words of a small int64 device tensor as a flat list: asynchronous copy into pinned memory + a polled event.  `.cpu()` parks the thread in
    the runtime's blocking wait, whose wake-up took ~0.25 ms longer than the GPU needed (measured: 0.35 ms in `.cpu()` per step for
    0.26 ms of queued kernels) — the step of a caller with fresh index tensors waits for exactly this read."""
    n = t.numel()
    bufs = getattr(_READ_BUFS, "bufs", None)
    if bufs is None:
        bufs = _READ_BUFS.bufs = {}
    dev = t.device
    got = bufs.get(dev)
    if got is None or got[0].numel() < n:
        got = bufs[dev] = (torch.empty(max(n, 64), dtype=torch.int64, pin_memory=True), torch.cuda.Event())
    host, ev = got
    host[:n].copy_(t.reshape(-1), non_blocking=True)
    ev.record(torch.cuda.current_stream(dev))
    while not ev.query():
        pass
    return host[:n].tolist()


def _fingerprint_applies(tensors) -> bool:
    if not FINGERPRINT or not all(t.is_cuda and t.dtype in (torch.int32, torch.int64) for t in tensors):
        return False
    total = sum(t.numel() for t in tensors)
    if total < FINGERPRINT_MIN or sum(t.numel() * t.element_size() for t in tensors) > INDEX_COPY_MAX_BYTES:
        return False
    return not torch.cuda.is_current_stream_capturing()


def _after_its_writer(core: _Core, stream) -> None:
    """Order `stream` behind the launches that wrote the core's fingerprint words and index copies (queued on another stream)."""
    ev = core.own.get("fp_event")
    if ev is not None and core.own.get("fp_stream") != stream.cuda_stream:
        stream.wait_event(ev)


def _launch_match(tensors, uniq):
    """Queue the pass that fingerprints `tensors` and compares them with the most recently used candidate's copy; nothing is read
    back.  Returns what _finish_match needs."""
    from . import _backend

    dev = tensors[0].device
    cur = torch.cuda.current_stream(dev)
    first = uniq[0]
    _after_its_writer(first, cur)
    STATS["fingerprints"] += 1
    out, _ = _backend.index_fingerprint_match(tensors, refs=first.own["index_copy"], hash=False)      # (compare only: 12 us less at C2)
    # ONE device-to-host copy: the new words and the fingerprints of the candidates whose words are not on the host yet
    missing = [c for c in uniq if "fp_host" not in c.own]
    for c in missing:
        _after_its_writer(c, cur)
    stacked = torch.cat([out.reshape(-1)] + [c.fp.reshape(-1) for c in missing]) if missing else out.reshape(-1)
    return _read_words_begin(stacked), missing, uniq      # (the copy to the host is queued HERE: before anything a Speculation queues)


def _finish_match(tensors, pending):
    """(adopted core or None, fingerprint words of `tensors` as a flat host list): one host read when the tensors equal the most
    recently used candidate; one more pass + read per candidate the fingerprint selects otherwise."""
    from . import _backend

    handle, missing, uniq = pending
    first = uniq[0]
    cur = torch.cuda.current_stream(tensors[0].device)
    k = len(tensors)
    flat = _read_words_end(handle)
    mine = flat[:3 * k]
    for i, c in enumerate(missing):
        c.own["fp_host"] = flat[3 * k + 2 * k * i:3 * k + 2 * k * (i + 1)]
    STATS["verified"] += 1
    if all(mine[3 * j + 2] == 0 for j in range(k)):
        return first, list(first.own["fp_host"])          # equal content: the candidate's fingerprint is this one's
    # other content (rare): now the fingerprint itself is needed — to look among the other candidates, and for the new cache entry
    out_h, _ = _backend.index_fingerprint_match(tensors)
    flat_h = _read_words(out_h.reshape(-1))
    words = [w for j in range(k) for w in flat_h[3 * j:3 * j + 2]]
    if first.own["fp_host"] == words:
        STATS["collisions"] += 1          # equal fingerprints, different content: exactly what the comparison is for
    for c in uniq[1:]:
        if c.own["fp_host"] != words:
            continue
        _after_its_writer(c, cur)
        out2, _ = _backend.index_fingerprint_match(tensors, refs=c.own["index_copy"])
        STATS["verified"] += 1
        if all(w == 0 for w in _read_words(out2[:, 2])):
            return c, words
        STATS["collisions"] += 1
    return None, words


def _candidates(geom):
    with _CACHE_LOCK:
        live = [c for c in _RECENT.values() if c.geom == geom] + [c for c in _CACHE.values() if c.geom == geom and "index_copy" in c.own]
    return live, list({id(c): c for c in reversed(live)}.values())          # most recently used first (then the recently orphaned)


class Speculation:
    """A cache miss whose answer has been ASKED FOR but not read: `candidate` is the most recently used pattern of the same geometry,
    the pass that compares the fresh index tensors with its copy is queued.  The caller may queue work that assumes the answer is
    "equal" (sparse_mm: the forward of the candidate's step plan) and must then call finish(): True = equal, the candidate's plans
    are adopted under the new key and what was queued is valid; False = other content (the cache entry of the new pattern exists
    now), what was queued must be DROPPED.  The host's reaction to the answer then overlaps the queued work instead of an idle GPU."""

    __slots__ = ("kind", "tensors", "shape", "pending", "candidate")

    def finish(self) -> bool:
        return _core_for(self.kind, self.tensors, self.shape, pending=self.pending) is self.candidate


def speculate_csr(A: torch.Tensor):
    """Speculation for a CSR tensor whose index tensors miss the cache while a live pattern of the same geometry exists, else None."""
    crow, col = A.crow_indices(), A.col_indices()
    tensors = (crow, col)
    key = _key("csr", tensors, A.shape)
    with _CACHE_LOCK:
        if key in _CACHE:
            return None
    if not _fingerprint_applies(tensors):
        return None
    geom = ("csr", tuple(A.shape)) + tuple((t.shape, t.dtype, t.device) for t in tensors)
    _, uniq = _candidates(geom)
    if not uniq:
        return None
    sp = Speculation()
    sp.kind, sp.tensors, sp.shape, sp.candidate = "csr", tensors, A.shape, uniq[0]
    sp.pending = _launch_match(tensors, uniq)
    return sp


def _core_for(kind: str, tensors, shape, pending=None) -> _Core:
    key = _key(kind, tensors, shape)
    with _CACHE_LOCK:
        core = _CACHE.get(key)
        if core is not None:
            _CACHE.move_to_end(key)
            if core.geom is not None:
                _FRESH.pop(core.geom, None)
            return core
    geom = (kind, tuple(shape)) + tuple((t.shape, t.dtype, t.device) for t in tensors)
    adopted, words, live = None, None, []
    applies = pending is not None or _fingerprint_applies(tensors)
    if applies:
        live, uniq = _candidates(geom)
        if pending is not None:            # (a Speculation: the comparison with pending's first candidate is queued already)
            live = live or list(pending[2])
            adopted, words = _finish_match(tensors, pending)
        elif uniq:
            adopted, words = _finish_match(tensors, _launch_match(tensors, uniq))
    if adopted is None:
        core = _Core()
        core.geom = geom
        if applies:
            # the core's own copy of the content it is built from (what later misses are compared with) — and its fingerprint, by the
            # same pass, unless the candidates' comparison has produced the words already
            from . import _backend

            cur = torch.cuda.current_stream(tensors[0].device)
            if words is None:
                STATS["fingerprints"] += 1
                out, copies = _backend.index_fingerprint_match(tensors, copy=True)
                core.fp = out[:, :2].contiguous()
            else:
                copies = [t.contiguous().clone() for t in tensors]
                core.own["fp_host"] = words
            core.own["index_copy"] = tuple(copies)
            ev = torch.cuda.Event()
            ev.record(cur)
            core.own["fp_event"], core.own["fp_stream"] = ev, cur.cuda_stream
    with _CACHE_LOCK:
        if adopted is not None:
            core = adopted
            _RECENT.pop(id(core), None)       # (it has a key again)
            STATS["adopted"] += 1
            _FRESH.pop(geom, None)
        elif applies and live:
            fresh = _FRESH[geom] = _FRESH.get(geom, 0) + 1
            if fresh >= FRESH_LIMIT:
                core.own["volatile"] = 0
                STATS["volatile"] += 1
        _CACHE[key] = core
        for t in tensors:
            weakref.finalize(t.untyped_storage(), _evict, key)
        while len(_CACHE) > _CACHE_MAX:
            _CACHE.popitem(last=False)
        if len(_CACHE) + len(_RECENT) > 1 and _cached_bytes() > _CACHE_MAX_BYTES:
            while _RECENT and _cached_bytes() > _CACHE_MAX_BYTES:
                _RECENT.popitem(last=False)
            while len(_CACHE) > 1 and _cached_bytes() > _CACHE_MAX_BYTES:
                _CACHE.popitem(last=False)
    return core


def _cached_bytes() -> int:
    return sum(c.nbytes() for c in {id(c): c for c in list(_CACHE.values()) + list(_RECENT.values())}.values())      # (adopted cores sit under several keys)


def clear_cache() -> None:
    with _CACHE_LOCK:
        _CACHE.clear()
        _RECENT.clear()
        _FRESH.clear()


def cache_stats():
    """(entries, derived bytes) held by the pattern cache."""
    with _CACHE_LOCK:
        return len(_CACHE), _cached_bytes()


def from_csr(A: torch.Tensor) -> RowGather:
    """Plan for a CSR tensor, 2-D or batched 3-D (torch's batched CSR: equal nnz per item)."""
    crow, col = A.crow_indices(), A.col_indices()
    return RowGather(crow, col, A.size(-2), A.size(-1), core=_core_for("csr", (crow, col), A.shape))


def from_coo_2d(indices: torch.Tensor, shape, coalesced: bool) -> RowGather:
    """Plan for 2-D COO indices (2, nnz).  Coalesced input is already row-sorted; otherwise the
    entries are visited in a stable row order through ``perm`` (duplicates stay separate)."""
    n, m = int(shape[-2]), int(shape[-1])
    core = _core_for("coo" + ("c" if coalesced else "u"), (indices,), shape)
    own = core.own
    if "crow" not in own:
        rows = indices[0]
        if coalesced:
            own["crow"] = torch._convert_indices_from_coo_to_csr(rows, n, out_int32=False)
        else:
            order = torch.argsort(rows, stable=True)
            own["crow"] = torch._convert_indices_from_coo_to_csr(rows[order].contiguous(), n, out_int32=False)
            own["col"] = indices[1][order].contiguous()
            own["perm"] = order
    if coalesced:
        return RowGather(own["crow"], indices[1].contiguous(), n, m, core=core)
    return RowGather(own["crow"], own["col"], n, m, perm=own["perm"], core=core)


def from_coo_batched(indices: torch.Tensor, shape) -> RowGather:
    """Plan for coalesced 3-D COO indices (3, nnz) as ONE block-diagonal 2-D pattern (b·n × b·m) — what the reference
    assembles for every batched input (sparse_matmul.py:151-153); items may differ in nnz.  Cached on the caller's
    index tensor (the flattened indices are derived data)."""
    b, n, m = (int(v) for v in shape)
    core = _core_for("coo3", (indices,), shape)
    own = core.own
    if "crow" not in own:
        own["crow"] = torch._convert_indices_from_coo_to_csr(indices[0] * n + indices[1], b * n, out_int32=False)
        own["col"] = (indices[0] * m + indices[2]).contiguous()
    return RowGather(own["crow"], own["col"], b * n, b * m, core=core)


def flat_of(g: RowGather) -> RowGather:
    """Batched CSR plan (b, n+1)/(b, nnz) → the 2-D block-diagonal plan (b·n × b·m) the reference assembles with
    ``sparse_block_diag`` (utils/utils.py:615-645): two vectorised adds, no sync; cached with the pattern."""
    if g.batch is None:
        return g
    if g.core.flat is None:
        crow, col, n, m = g.crow, g.col, g.n_rows, g.n_cols
        b, nnz = col.shape
        idt = crow.dtype
        if idt == torch.int32 and max(b * nnz, b * m, b * n) >= 2**31:
            idt = torch.int64
        item = torch.arange(b, device=col.device, dtype=idt).unsqueeze(1)
        flat_crow = torch.cat(((crow[:, :-1].to(idt) + item * nnz).reshape(-1),
                               torch.tensor([b * nnz], dtype=idt, device=col.device)))
        flat_col = (col.to(idt) + item * m).reshape(-1)
        flat_perm = None if g.perm is None else (g.perm.to(idt) + item * nnz).reshape(-1)
        g.core.flat = RowGather(flat_crow, flat_col, b * n, b * m, perm=flat_perm)
    return g.core.flat


def flat_block_diag(crow: torch.Tensor, col: torch.Tensor, n: int, m: int) -> RowGather:
    """Block-diagonal 2-D plan of batched CSR index arrays (see flat_of)."""
    b = col.size(0)
    g = RowGather(crow, col, n, m, core=_core_for("csr", (crow, col), (b, n, m)))
    return flat_of(g)
