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

import threading
from collections import OrderedDict
from typing import Optional, Tuple

import torch


class RowGather:
    """Row-gather structure of a (batched) sparse matrix on the device.

    crow: (n_rows+1,) or (b, n_rows+1); col: (nnz,) or (b, nnz); perm: optional, same shape as
    col, position of each entry in the owner's value array (None = identity).
    """

    __slots__ = ("crow", "col", "perm", "n_rows", "n_cols", "batch", "_t", "_has_diag", "_rows", "_tiles", "_blocks",
                 "_packs")

    def __init__(self, crow, col, n_rows, n_cols, perm=None):
        self.crow, self.col, self.perm = crow, col, perm
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.batch = crow.size(0) if crow.dim() == 2 else None
        self._t: Optional[RowGather] = None
        self._has_diag: Optional[bool] = None
        self._rows = None
        self._tiles = {}
        self._blocks = {}
        self._packs = {}

    def rowpack_plan(self, rows_per_block: int, limits):
        """Plan for the row-pair gather kernels (csrc/rowpack_impl.h), None when the pattern does not qualify or
        profit; cached per workgroup height.  See `build_rowpack_plan`."""
        key = (rows_per_block, tuple(limits))
        if key not in self._packs:
            plan = None
            if ENABLE_BRICKS and self.perm is not None:
                lat = detect_lattice(self)
                order = brick_pair_order(self.n_rows, lat, rows_per_block // 2, self.crow.device) if lat else None
                if order is not None:
                    plan = build_rowpack_plan(self, rows_per_block, limits, pair_order=order, lattice=lat)
            self._packs[key] = plan if plan is not None else build_rowpack_plan(self, rows_per_block, limits)
        return self._packs[key]

    def block_plan(self, rows_per_block: int, row_bytes: int, limits):
        """Plan for the workgroup-tiled kernels (csrc/blocktile_impl.h), None when the pattern does not
        qualify or profit; cached per block height / dense row size.  See `build_block_plan`."""
        key = (rows_per_block, row_bytes, tuple(limits))
        if key not in self._blocks:
            self._blocks[key] = build_block_plan(self, rows_per_block, row_bytes, limits)
        return self._blocks[key]

    def tiles(self, rows_per_task: int, max_distinct: int, max_entries: int):
        """Plan for the wave-pipelined LDS-tiled kernels (None when the pattern does not qualify or
        profit); cached per task height.  See `build_tile_plan`."""
        key = (rows_per_task, max_distinct, max_entries)
        if key not in self._tiles:
            self._tiles[key] = build_tile_plan(self, rows_per_task, max_distinct, max_entries)
        return self._tiles[key]

    @property
    def nnz(self) -> int:
        return self.col.size(-1)

    def row_indices(self) -> torch.Tensor:
        """Expanded row index per stored entry (same shape/dtype as col); cached."""
        if self._rows is None:
            n = self.n_rows
            ar = torch.arange(n, dtype=self.col.dtype, device=self.col.device)
            if self.batch is None:
                self._rows = torch.repeat_interleave(ar, self.crow[1:] - self.crow[:-1], output_size=self.nnz)
            else:
                counts = (self.crow[:, 1:] - self.crow[:, :-1]).reshape(-1)
                rows = torch.repeat_interleave(ar.repeat(self.batch), counts, output_size=self.batch * self.nnz)
                self._rows = rows.view(self.batch, self.nnz)
        return self._rows

    @property
    def transposed(self) -> "RowGather":
        """Row-gather structure of Aᵀ whose ``perm`` indexes A's value array."""
        if self._t is None:
            self._t = _transpose(self)
        return self._t

    @property
    def has_diagonal(self) -> bool:
        if self._has_diag is None:
            self._has_diag = bool(torch.any(self.row_indices() == self.col))
        return self._has_diag


class TilePlan:
    """Per-task column dictionary consumed by tsgu_csr_*_wavetile (layout: include/tsgu_hip.h)."""

    __slots__ = ("tmeta", "tile_cols", "lidx", "reuse", "max_distinct", "max_entries", "nnz")

    def __init__(self, tmeta, tile_cols, lidx, reuse, max_distinct, max_entries, nnz):
        self.tmeta, self.tile_cols, self.lidx = tmeta, tile_cols, lidx
        self.reuse, self.max_distinct, self.max_entries, self.nnz = reuse, max_distinct, max_entries, nnz


_TILE_MIN_REUSE = 1.5   # average entries per distinct column inside a task
_TILE_MAX_PADDING = 2.0  # padded / real size of the per-task tables


def build_tile_plan(g: RowGather, rows_per_task: int, cap_distinct: int, cap_entries: int):
    """Distinct columns per task of `rows_per_task` consecutive rows + 8-bit local indices, in the
    fixed-stride, 16-byte aligned layout the wave-pipelined kernels load with wide accesses.

    One sort of (task, column) keys per pattern (tens of ms at 27e6 entries), amortised over every
    later forward/backward on the same pattern.  Returns None when a task exceeds the kernel limits,
    the pattern has too little column reuse, or padding would waste memory (ragged patterns)."""
    if g.batch is not None or g.n_rows == 0 or not (4 <= g.nnz < 2**31):
        return None
    n, m, nnz = g.n_rows, g.n_cols, g.nnz
    dev = g.crow.device
    ntask = (n + rows_per_task - 1) // rows_per_task
    e0 = g.crow[torch.arange(0, n, rows_per_task, device=dev)].to(torch.int64)
    ne = torch.cat((e0[1:], g.crow[-1:].to(torch.int64))) - e0
    if int(ne.max()) > cap_entries or ntask * cap_entries > _TILE_MAX_PADDING * nnz:
        return None
    task = g.row_indices().to(torch.int64) // rows_per_task
    key = task * m + g.col.to(torch.int64)
    uniq, inv = torch.unique(key, return_inverse=True)
    total = uniq.numel()
    reuse = nnz / max(total, 1)
    if reuse < _TILE_MIN_REUSE:
        return None
    utask = uniq // m
    cnt = torch.bincount(utask, minlength=ntask)
    top = int(cnt.max())
    if top > cap_distinct or top > 256 or ntask * cap_distinct > _TILE_MAX_PADDING * total:
        return None
    first = torch.cumsum(cnt, 0) - cnt                       # start of each task's run in `uniq`
    # padded column table: position d of task t reads uniq[first[t] + min(d, cnt[t]-1)]
    d = torch.arange(cap_distinct, device=dev).unsqueeze(0)
    src = first.unsqueeze(1) + torch.minimum(d, (cnt - 1).clamp_min(0).unsqueeze(1))
    cols = (uniq - utask * m).to(torch.int32)
    tile_cols = cols[src.clamp_max(total - 1)].contiguous()  # (ntask, cap_distinct)
    # padded local-index table: entry e of task t (task order) -> position inside the task's run
    local = (inv - first[task]).to(torch.uint8)
    pos = torch.arange(nnz, device=dev, dtype=torch.int64) - e0[task]
    lidx = torch.zeros((ntask, cap_entries), dtype=torch.uint8, device=dev)
    lidx[task, pos] = local
    tmeta = torch.stack((e0, ne), dim=1).to(torch.int32).contiguous()
    return TilePlan(tmeta, tile_cols, lidx, reuse, top, int(ne.max()), nnz)


class BlockPlan:
    """Per-block dictionary of distinct dense rows + packed entry words for tsgu_csr_*_blocktile
    (layout: include/tsgu_hip.h)."""

    __slots__ = ("ndist", "trow", "ent", "sperm", "capd", "ecap", "rpb", "reuse", "nnz")

    def __init__(self, ndist, trow, ent, sperm, capd, ecap, rpb, reuse, nnz):
        self.ndist, self.trow, self.ent, self.sperm = ndist, trow, ent, sperm
        self.capd, self.ecap, self.rpb, self.reuse, self.nnz = capd, ecap, rpb, reuse, nnz


def build_block_plan(g: RowGather, rpb: int, row_bytes: int, limits):
    """Blocks of `rpb` consecutive rows: distinct column indices per block (`trow`, padded to `capd`), a 16-bit
    local index per stored entry and — for plans that address the owner's values through `perm`
    (transposed / un-coalesced) — the permutation sorted inside each block (`sperm`) with each entry's slot in
    that order (`ent = lidx | slot << 16`).  Two device sorts per pattern, amortised over every later call.
    `limits` = (distinct_multiple, max_distinct, max_entries, lds_budget_bytes) from tsgu_blocktile_limits."""
    mult, max_distinct, max_entries, lds_budget = limits
    if g.batch is not None or g.n_rows == 0 or not (1 <= g.nnz < 2**31):
        return None
    n, m, nnz = g.n_rows, g.n_cols, g.nnz
    dev = g.crow.device
    nb = (n + rpb - 1) // rpb
    e0 = g.crow[torch.arange(0, n, rpb, device=dev)].to(torch.int64)
    ne = torch.cat((e0[1:], g.crow[-1:].to(torch.int64))) - e0
    ecap = (int(ne.max()) + 255) // 256 * 256
    if ecap == 0 or ecap > max_entries:
        return None
    blk = g.row_indices().to(torch.int64) // rpb
    uniq, inv = torch.unique(blk * m + g.col.to(torch.int64), return_inverse=True)
    total = uniq.numel()
    reuse = nnz / max(total, 1)
    if reuse < _TILE_MIN_REUSE:
        return None
    ublk = uniq // m
    cnt = torch.bincount(ublk, minlength=nb)
    capd = (int(cnt.max()) + mult - 1) // mult * mult
    if capd > max_distinct or capd * row_bytes + ecap * 8 > lds_budget or nb * capd > _TILE_MAX_PADDING * total + 4096:
        return None
    first = torch.cumsum(cnt, 0) - cnt
    d = torch.arange(capd, device=dev).unsqueeze(0)
    src = first.unsqueeze(1) + torch.minimum(d, (cnt - 1).clamp_min(0).unsqueeze(1))
    trow = (uniq - ublk * m).to(torch.int32)[src.clamp_max(total - 1)].contiguous()
    ent = (inv - first[blk]).to(torch.int64)
    sperm = None
    if g.perm is not None:
        # entries are stored block after block, so sorting (block, perm) permutes inside each block only
        order = torch.argsort(blk * nnz + g.perm.to(torch.int64))
        sperm = g.perm[order].to(torch.int32).contiguous()
        slot = torch.empty(nnz, dtype=torch.int64, device=dev)
        slot[order] = torch.arange(nnz, device=dev, dtype=torch.int64) - e0[blk[order]]
        ent = ent | (slot << 16)
    ent = ent.to(torch.int32).contiguous()  # bit pattern of the uint32 word (slot < 2048)
    return BlockPlan(cnt.to(torch.int32).contiguous(), trow, ent, sperm, capd, ecap, rpb, reuse, nnz)


class RowPackPlan:
    """Union-of-columns walk for pairs of consecutive rows, consumed by tsgu_csr_*_rowpack (layout: include/tsgu_hip.h)."""

    __slots__ = ("uptr", "ucol", "upos", "sperm", "order", "vpair", "eptr", "nblocks", "ecap", "ucap", "rpb", "reuse",
                 "nnz", "lattice")

    def __init__(self, uptr, ucol, upos, sperm, ecap, ucap, rpb, reuse, nnz, order=None, vpair=None, eptr=None,
                 nblocks=0, lattice=None):
        self.uptr, self.ucol, self.upos, self.sperm, self.order = uptr, ucol, upos, sperm, order
        self.vpair, self.eptr, self.nblocks, self.lattice = vpair, eptr, nblocks, lattice
        self.ecap, self.ucap, self.rpb, self.reuse, self.nnz = ecap, ucap, rpb, reuse, nnz


_PACK_MIN_REUSE = 1.2   # stored entries per union entry (2.0 = both rows of every pair share all columns)
_PACK_ABSENT = 0x8000
ENABLE_BRICKS = True    # lattice patterns: permuted walks own 3-D bricks of row pairs (see brick_pair_order)


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


def build_rowpack_plan(g: RowGather, rpb: int, limits, pair_order=None, lattice=None):
    """Rows 2q and 2q+1 walk the sorted union of their column sets: `ucol` per union entry, `upos` = two 16-bit slots
    (one per row; bit 15 = this row has no entry there) into the value slice a workgroup of `rpb` rows stages.  For
    plans addressed through `perm` the slice is staged in the order of the permutation sorted inside the workgroup
    (`sperm`), otherwise in stored order.  `pair_order` (permuted plans only) assigns row pairs to lane-group slots
    (see brick_pair_order); None = consecutive.  `limits` = (max_entries, max_union, lds_budget_bytes)."""
    max_entries, max_union, lds_budget = limits
    if g.batch is not None or g.n_rows == 0 or not (1 <= g.nnz < 2**31):
        return None
    n, m, nnz = g.n_rows, g.n_cols, g.nnz
    dev = g.crow.device
    gpb = rpb // 2
    npairs = (n + 1) // 2
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
    vp = slot_of[rows // 2]                       # lane-group slot of every stored entry
    blk = vp // gpb
    ne = torch.bincount(blk, minlength=nb)
    ecap = max((int(ne.max()) + 255) // 256 * 256, 256)
    if ecap > max_entries or ecap >= _PACK_ABSENT:
        return None
    eptr = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    eptr[1:] = torch.cumsum(ne, 0)
    uniq, inv = torch.unique(vp * m + g.col.to(torch.int64), return_inverse=True)  # sorted: (slot, column) ascending
    nu = uniq.numel()
    reuse = nnz / max(nu, 1)
    if reuse < _PACK_MIN_REUSE:
        return None
    uslot = uniq // m
    uptr = torch.zeros(nslots + 1, dtype=torch.int64, device=dev)
    uptr[1:] = torch.cumsum(torch.bincount(uslot, minlength=nslots), 0)
    ub = uptr[torch.arange(0, nslots + 1, gpb, device=dev)]
    ucap = max((int((ub[1:] - ub[:-1]).max()) + 255) // 256 * 256, 256)
    if ucap > max_union or ucap * (4 if g.perm is None else 8) + ecap * 4 > lds_budget or (g.perm is None and m >= 2**30):
        return None
    k = torch.arange(nnz, device=dev, dtype=torch.int64)
    sperm = None
    if g.perm is None:
        slot = k - eptr[blk]                      # natural order: a workgroup's entries are one contiguous range
    else:
        order = torch.argsort(blk * nnz + g.perm.to(torch.int64))
        sperm = g.perm[order].to(torch.int32).contiguous()
        slot = torch.empty(nnz, dtype=torch.int64, device=dev)
        slot[order] = k - eptr[blk[order]]
    to_i32 = lambda w: torch.where(w >= 2**31, w - 2**32, w).to(torch.int32).contiguous()  # noqa: E731  (uint32 bit pattern)
    ucol64 = uniq - uslot * m
    if g.perm is None:
        # stored order: a row's slots are consecutive, the record only carries the ownership bits (30: row 2q, 31: 2q+1)
        own = torch.zeros((2, nu), dtype=torch.int64, device=dev)
        own[rows % 2, inv] = 1
        ucol = to_i32(ucol64 | (own[0] << 30) | (own[1] << 31))
        word = None
    else:
        half = torch.full((2, nu), _PACK_ABSENT, dtype=torch.int64, device=dev)
        half[rows % 2, inv] = slot
        word = to_i32(half[0] | (half[1] << 16))
        ucol = ucol64.to(torch.int32).contiguous()
    plan = RowPackPlan(uptr.to(torch.int32).contiguous(), ucol, word, sperm, ecap, ucap, rpb, reuse, nnz)
    if not natural:
        plan.vpair = pair_order.to(torch.int32).contiguous()
        plan.eptr = eptr.to(torch.int32).contiguous()
        plan.nblocks = nb
        plan.lattice = lattice
    return plan


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

_CACHE: "OrderedDict[tuple, Tuple[tuple, RowGather]]" = OrderedDict()
_CACHE_LOCK = threading.Lock()  # backward runs on autograd threads
_CACHE_MAX = 16


def _key(kind: str, tensors, shape) -> tuple:
    return (kind, tuple(shape)) + tuple((t.data_ptr(), t._version, tuple(t.shape), t.dtype, str(t.device)) for t in tensors)


def _cached(kind, tensors, shape, build) -> RowGather:
    key = _key(kind, tensors, shape)
    with _CACHE_LOCK:
        hit = _CACHE.get(key)
        if hit is not None:
            _CACHE.move_to_end(key)
            return hit[1]
    plan = build()
    with _CACHE_LOCK:
        # keep the index tensors alive with the entry so their addresses cannot be recycled
        _CACHE[key] = (tuple(tensors), plan)
        while len(_CACHE) > _CACHE_MAX:
            _CACHE.popitem(last=False)
    return plan


def clear_cache() -> None:
    with _CACHE_LOCK:
        _CACHE.clear()


def from_csr(A: torch.Tensor) -> RowGather:
    """Plan for a CSR tensor, 2-D or batched 3-D (torch's batched CSR: equal nnz per item)."""
    crow, col = A.crow_indices(), A.col_indices()
    return _cached("csr", (crow, col), A.shape, lambda: RowGather(crow, col, A.size(-2), A.size(-1)))


def from_coo_2d(indices: torch.Tensor, shape, coalesced: bool) -> RowGather:
    """Plan for 2-D COO indices (2, nnz).  Coalesced input is already row-sorted; otherwise the
    entries are visited in a stable row order through ``perm`` (duplicates stay separate)."""

    def build():
        n, m = int(shape[-2]), int(shape[-1])
        rows, cols = indices[0], indices[1]
        if coalesced:
            crow = torch._convert_indices_from_coo_to_csr(rows, n, out_int32=False)
            return RowGather(crow, cols.contiguous(), n, m)
        order = torch.argsort(rows, stable=True)
        crow = torch._convert_indices_from_coo_to_csr(rows[order].contiguous(), n, out_int32=False)
        return RowGather(crow, cols[order].contiguous(), n, m, perm=order)

    return _cached("coo" + ("c" if coalesced else "u"), (indices,), shape, build)


def flat_block_diag(crow: torch.Tensor, col: torch.Tensor, n: int, m: int) -> RowGather:
    """Batched CSR arrays (b, n+1)/(b, nnz) → the 2-D block-diagonal plan (b·n × b·m) the reference
    assembles with ``sparse_block_diag`` (utils/utils.py:615-645): two vectorised adds, no sync."""

    def build():
        b, nnz = col.shape
        idt = crow.dtype
        if idt == torch.int32 and max(b * nnz, b * m) >= 2**31:
            idt = torch.int64
        item = torch.arange(b, device=col.device, dtype=idt).unsqueeze(1)
        flat_crow = torch.cat(((crow[:, :-1].to(idt) + item * nnz).reshape(-1),
                               torch.tensor([b * nnz], dtype=idt, device=col.device)))
        flat_col = (col.to(idt) + item * m).reshape(-1)
        return RowGather(flat_crow, flat_col, b * n, b * m)

    return _cached("csrflat", (crow, col), (n, m), build)
