"""Lattice plans: the description of a stencil-like sparsity pattern that the plane-sweep kernels
(csrc/lattice_impl.h, C ABI ``tsgu_csr_spmm_lattice`` / ``tsgu_csr_sddmm_lattice``) walk.

A pattern qualifies when its rows are the points of a row-major lattice, ``row = ((item·nx + x)·ny + y)·nz + z``, and
every stored entry couples a point with a neighbour at a displacement ``(dx, dy, dz)``, ``|dx| <= 1``, ``|dy|, |dz| <= 2``
(periodic wrap allowed; x wraps inside an item of a batched / block-diagonal problem).  That covers what the reference's
``PairwiseEncoder`` produces (encoders/pairwise_encoder.py) and the stencil / Laplacian matrices of its benchmarks
(7- and 27-point, periodic or truncated at the boundary, triangular parts of them, 2-D 5- and 9-point).  Rows with the
same displacement sequence share a *class*; the plan is the class of every row (one byte), one displacement table per
class and — for the transposed walk — the position of every entry inside its source row.  Everything else (the LDS
record tables of a launch configuration) is derived from these few hundred numbers on the host.

Nothing here touches the values: the plan survives value updates (the training-loop case), like every other plan.
"""

from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional, Tuple

import torch

MAX_CLASSES = 255
MAX_RADIUS = 2
MAX_LEN = 32
PAD_LO = PAD_HI = 0x7FF00     # bytes: reads this far beyond the row's own LDS position are beyond the allocation and return 0
_MODE_SPMM, _MODE_SDDMM, _MODE_SPMMT = 0, 1, 2
PACKED_T = False    # one-word records for the transposed walk (must match kPacked in csrc/lattice_impl.h; measured slower)
_CU_COUNT: Dict[int, int] = {}


def num_cu(device=None) -> int:
    """Compute units of `device` (cached per device; 256 on an MI355X — and when there is no GPU to ask: CPU-side tests of the
    configuration ranking)."""
    idx = -1
    if device is not None and getattr(device, "type", "cpu") == "cuda":
        idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _CU_COUNT:
        n = 256
        if idx >= 0:
            try:
                # (an attribute query: torch.cuda.get_device_properties reads the whole property structure — 117 ms of a process'
                # first sparse_mm step went there)
                from . import _backend

                n = int(_backend.device_cu_count(idx)) or 256
            except Exception:       # noqa: BLE001 - the count only steers a launch-configuration choice
                n = 256
        _CU_COUNT[idx] = n
    return _CU_COUNT[idx]


class LatticePlan:
    """Geometry + row classes of one pattern (kind 0: walked in stored order; kind 1: a transposed pattern whose values
    live in the source rows of the owner's value array)."""

    __slots__ = ("kind", "nb", "nx", "ny", "nz", "ry", "rz", "ncls", "recw", "uniform_len", "codes", "ksrc", "lens",
                 "lens_host", "rcls", "rstart", "n_rows", "nnz", "box", "_cfg", "_march")

    def __init__(self):
        self._cfg: Dict[tuple, "LatticeConfig"] = {}
        self.box = None              # (mask, periodic bits) when the pattern meets the plane-march condition (checked per row)
        self._march = False          # False: not derived yet; None: not a box stencil; else MarchTables

    def plan_bytes(self) -> int:
        total = 0
        for t in (self.lens, self.rcls, self.rstart):
            if t is not None:
                total += t.numel() * t.element_size()
        for cfg in self._cfg.values():
            if cfg is not None:
                total += cfg.rec.numel() * cfg.rec.element_size() + cfg.wlist.numel()
        return total


class LatticeConfig:
    """One launch configuration of a plan: tile, segments, workgroup size and the record tables built for them."""

    __slots__ = ("ty", "tz", "nseg", "threads", "rec", "lds_bytes", "struct", "struct_addr", "wlist", "nloc", "ring", "cpl", "uses", "tuned")


SAMPLE_ROWS = 256


def _clusters_to_dims(pos, n: int) -> Optional[Tuple[int, int]]:
    """(nz, ny·nz) from the sorted frequent |col − row| offsets `pos` (see detect_dims)."""
    if not pos:
        return None
    clusters = [[pos[0]]]
    reach = max(pos[0], MAX_RADIUS)
    for o in pos[1:]:
        if o - clusters[-1][-1] <= reach:
            clusters[-1].append(o)
        else:
            reach = clusters[-1][-1] + MAX_RADIUS
            clusters.append([o])
    if clusters[0][0] > MAX_RADIUS:
        clusters.insert(0, [])
    if len(clusters) not in (2, 3) or (clusters[0] and clusters[0][-1] > MAX_RADIUS):
        return None
    lo, hi = clusters[1][0], clusters[1][-1]
    mid = (lo + hi) // 2
    cands = sorted((c for c in range(max(hi - MAX_RADIUS, 2), lo + MAX_RADIUS + 1) if n % c == 0), key=lambda c: abs(c - mid))
    if not cands:
        return None
    nz = cands[0]
    if len(clusters) == 2:
        return nz, nz
    lo, hi = clusters[2][0], clusters[2][-1]
    mid = (lo + hi) // 2
    first = (max(lo - MAX_RADIUS, 2 * nz) + nz - 1) // nz * nz
    cands = sorted((c for c in range(first, hi + MAX_RADIUS + 1, nz) if n % c == 0), key=lambda c: abs(c - mid))
    if not cands:
        return None
    return nz, cands[0]


def _sample_dims(g) -> Optional[Tuple[int, int, int, int]]:
    """(nb, nx, ny, nz) guessed on the HOST from small samples of rows: SAMPLE_ROWS rows from the middle of the matrix give
    the strides (interior rows: no wrap-around offsets), the first SAMPLE_ROWS rows (the x = 0 plane of the first item) give
    the x period.  Only a proposal: the row kernels check every entry.  Plain Python on two short lists — on a fresh
    process every distinct library routine (a torch reduction, numpy's unique) first has to be paged in, which costs
    more than this whole function."""
    from collections import Counter

    n = g.n_rows
    m = min(n, SAMPLE_ROWS)
    r0 = max(0, n // 2 - m // 2)

    def sample(first):
        ptr = g.crow[first:first + m + 1].tolist()
        cols = g.col[ptr[0]:ptr[-1]].tolist()
        return ptr, cols

    found = None
    # the strides from interior rows: the middle of the matrix, or (block-diagonal batches put an item boundary there) two
    # other places
    for first in dict.fromkeys((r0, max(0, n // 3 - m // 2), max(0, min(n - m, (2 * n) // 3 + 17 * m)))):
        ptr, cols = sample(first)
        if not cols:
            continue
        cnt = Counter()
        base = ptr[0]
        for i in range(m):
            r = first + i
            cnt.update(abs(c - r) for c in cols[ptr[i] - base:ptr[i + 1] - base])
        found = _clusters_to_dims(sorted(o for o, k in cnt.items() if k * 2 > m and o > 0), n)
        if found is not None and n % found[1] == 0 and n % found[0] == 0:
            break
        found = None
    if found is None:
        return None
    nz, d2 = found
    planes = n // d2
    ptr, cols = sample(0)
    base, mx = ptr[0], 0
    for i in range(m):
        px = i // d2
        for c in cols[ptr[i] - base:ptr[i + 1] - base]:
            mx = max(mx, abs(c // d2 - px))
    nx = planes if mx <= 1 else mx + 1
    if nx < 1 or planes % nx:
        return None
    return planes // nx, nx, d2 // nz, nz


def build_lattice_plan_hip(g, be, forward: Optional[LatticePlan] = None, dims=None) -> Optional[LatticePlan]:
    """The same plan as `build_lattice_plan`, with the per-entry work done by the row kernels of csrc/lattice_plan.hip
    (two passes over the pattern, nothing sorted: the distinct row hashes meet in a small device hash table) instead of
    ~40 tensor ops over the entries.
    `forward` None: plan of the stored-order walk of `g`.  `forward` = that plan: plan of the TRANSPOSED walk of the same
    `g` — found by searching the neighbour rows, the transposed pattern is never built."""
    import numpy as np

    if g.batch is not None or g.perm is not None or g.n_rows != g.n_cols or g.n_rows < 8 or not (1 <= g.nnz < 2**31):
        return None
    n = g.n_rows
    dev = g.crow.device
    kind = 0 if forward is None else 1
    if kind == 1:
        dims = (forward.nb, forward.nx, forward.ny, forward.nz)
        codes = forward.codes
        disp = torch.unique(codes[codes >= 0]).to(torch.uint8).to(dev)
    else:
        disp = None
        if dims is None:
            dims = _sample_dims(g)
        if dims is None:
            return None
    crow, col = g.crow.contiguous(), g.col.contiguous()
    slots = be.load_library().tsgu_lattice_slots()
    # one small work buffer: status[8] int32 | trep[slots] int32 | thash[slots] int64, pre-set by one copy from the host
    NST = 8
    init = np.empty(NST + slots + 2 * slots, dtype=np.int32)
    init[:NST] = 0
    init[NST:NST + slots] = np.iinfo(np.int32).max
    init[NST + slots:].view(np.int64)[:] = np.iinfo(np.int64).min
    work = torch.from_numpy(init).to(dev)
    status, trep, thash = work[:NST], work[NST:NST + slots], work[NST + slots:].view(torch.int64)
    slot = torch.empty(n, dtype=torch.int16, device=dev)
    be.lattice_rows(crow, col, dims, status, slot, thash=thash, trep=trep, disp=disp)
    host = work.cpu().numpy()
    bad, ry, rz, maxlen = (int(v) for v in host[:4])
    if bad or maxlen > MAX_LEN or maxlen < 1:
        return None
    hh = host[NST + slots:].view(np.int64)
    used = np.nonzero(hh != np.iinfo(np.int64).min)[0]
    ncls = used.size
    if ncls > MAX_CLASSES or ncls == 0:
        return None
    order = used[np.argsort(hh[used], kind="stable")]          # classes numbered by ascending hash, like torch.unique
    remap = np.zeros(slots, dtype=np.uint8)
    remap[order] = np.arange(ncls, dtype=np.uint8)
    rep_host = host[NST:NST + slots][order].astype(np.int64)
    rep = torch.from_numpy(rep_host).to(dev)
    table = torch.empty((ncls, 32), dtype=torch.int32, device=dev)
    be.lattice_row_codes(crow, col, dims, rep, table, disp=disp)
    tab_host = table.cpu()                                   # [classes][32] words: the rest of the class bookkeeping is host work
    lens = torch.from_numpy((tab_host.numpy() >= 0).sum(1).astype(np.uint8)).to(dev)
    rcls = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    status.zero_()
    # the plane-march condition (stored-order walk): proposed from the class representatives, checked per row by pass 2
    box = box_candidate(tab_host.numpy(), rep_host, dims, ry, rz) if kind == 0 else None
    be.lattice_rows(crow, col, dims, status, slot, remap=torch.from_numpy(remap).to(dev), ctable=table, lens=lens, rcls=rcls, disp=disp,
                    box_mask=box[0] if box else 0, periodic=box[1] if box else 0)
    st2 = status.cpu()
    if int(st2[0]) != 0:                                     # exact check: hash collisions must not pass
        return None
    if box is not None and int(st2[4]) != 0:
        box = None
    tab = tab_host.to(torch.int64)
    cl = (tab >= 0).sum(1)
    if kind == 0:
        uniform = int(cl[0]) if bool((cl == cl[0]).all()) else 0
        if uniform:
            rstart = torch.zeros(4, dtype=torch.int32, device=dev)
        else:
            rstart = (crow if crow.dtype == torch.int32 else crow.to(torch.int32)).clone()
        if dims[2] == 1:
            ry = 0
    else:
        uniform, rstart, ry, rz = forward.uniform_len, forward.rstart, forward.ry, forward.rz
    nb, nx, ny, nz = dims
    recw = (int(cl.max()) + 3) // 4 * 4
    tab = tab[:, :recw].contiguous()
    plan = LatticePlan()
    plan.kind, plan.nb, plan.nx, plan.ny, plan.nz, plan.ry, plan.rz = kind, nb, nx, ny, nz, ry, rz
    plan.ncls, plan.recw, plan.uniform_len = ncls, recw, uniform
    plan.n_rows, plan.nnz = n, g.nnz
    if kind == 1:
        plan.ksrc = torch.where(tab >= 0, tab % 32, tab)
        plan.codes = torch.where(tab >= 0, torch.div(tab, 32, rounding_mode="floor"), tab)
    else:
        plan.codes, plan.ksrc = tab, None
    plan.lens_host = cl
    plan.lens = lens
    plan.rcls = rcls
    plan.rstart = rstart
    plan.box = box
    return plan


def box_candidate(codes, rep_rows, dims, ry: int, rz: int):
    """(mask, periodic) of the plane-march condition proposed for a stored-order pattern, or None when it cannot hold:
    `codes` [classes][32] displacement codes of the class representatives `rep_rows` (-1 beyond a row).  mask = the displacements
    that occur (27 bits, bit (dx+1)·9 + (dy+1)·3 + dz+1); periodic bit 0 / 1 / 2 = some representative reaches across the x / y / z
    face it sits at.  Whether EVERY row holds exactly the displacements of `mask` whose neighbour exists is for the row kernel."""
    nb, nx, ny, nz = (int(v) for v in dims)
    if min(nx, ny, nz) < 3 or ry > 1 or rz > 1 or len(rep_rows) > MARCH_MAX_CLASSES:
        return None
    mask, periodic = 0, 0
    for row, seq in zip(rep_rows, codes):
        row = int(row)
        z, y, x = row % nz, (row // nz) % ny, (row // (ny * nz)) % nx
        for code in seq:
            code = int(code)
            if code < 0:
                continue
            dz, dy, dx = code % 5 - 2, (code // 5) % 5 - 2, code // 25 - 1
            if abs(dy) > 1 or abs(dz) > 1:
                return None
            mask |= 1 << ((dx + 1) * 9 + (dy + 1) * 3 + dz + 1)
            if not 0 <= x + dx < nx:
                periodic |= 1
            if not 0 <= y + dy < ny:
                periodic |= 2
            if not 0 <= z + dz < nz:
                periodic |= 4
    return (mask, periodic) if mask else None


def workgroup_classes_hip(plan: LatticePlan, ty: int, tz: int, nseg: int, be) -> torch.Tensor:
    """`workgroup_classes` by one kernel (a 256-bit class set per workgroup) + a few host operations on the sets."""
    import numpy as np

    nb, nx, ny, nz = plan.nb, plan.nx, plan.ny, plan.nz
    nblocks = nb * nseg * -(-ny // ty) * -(-nz // tz)
    dev = plan.rcls.device
    mask = torch.zeros((nblocks, 4), dtype=torch.int64, device=dev)
    be.lattice_block_classes(plan.rcls, plan.n_rows, (nb, nx, ny, nz), ty, tz, nseg, mask)
    bits = np.unpackbits(mask.cpu().numpy().view(np.uint8).reshape(nblocks, 32), axis=1, bitorder="little")   # [nblocks][256]
    nloc = int(bits.sum(1).max())
    # stable argsort puts the set bits first (as ascending class ids)
    idx = np.argsort(1 - bits, axis=1, kind="stable")[:, :nloc].astype(np.uint8)
    wl = np.where(np.take_along_axis(bits, idx.astype(np.int64), axis=1) == 1, idx, np.uint8(0xFF))
    return torch.from_numpy(np.ascontiguousarray(wl)).to(dev)


def records(plan: LatticePlan, ty: int, tz: int, row_bytes: int, slot_bytes: int, ring: int = 4, elem_bytes: int = 4) -> torch.Tensor:
    """Record tables of `plan` for a ty × tz tile (host tensor, int32): [ring][ncls][recw] byte offsets for kind 0,
    [ring][ncls][recw][2] = (dense-row offset, value offset) for kind 1.  Entry k of a row of class c gathers the LDS row
    `rec` bytes from the row's own position in the halo tile, in the ring slot (phase + dx) % ring — see include/tsgu_hip.h."""
    hz = tz + 2 * plan.rz
    hr = (ty + 2 * plan.ry) * hz
    codes = plan.codes
    valid = codes >= 0
    c = torch.where(valid, codes, torch.zeros_like(codes))
    dz = c % 5 - 2
    dy = torch.div(c, 5, rounding_mode="floor") % 5 - 2
    dx = torch.div(c, 25, rounding_mode="floor") - 1
    out = []
    for ph in range(ring):
        slot = (ph + dx) % ring
        lo = slot * (hr * row_bytes) + (dy * hz + dz) * row_bytes
        lo = torch.where(valid, lo, torch.full_like(lo, PAD_LO))
        if plan.kind == 0:
            out.append(lo.to(torch.int32))
        else:
            k = torch.where(valid, plan.ksrc, torch.zeros_like(plan.ksrc))
            if PACKED_T and row_bytes % 128 == 0:
                # the value ring has the pitch of the dense ring: one word, value index in the low 7 bits
                assert slot_bytes == row_bytes
                out.append(torch.where(valid, lo + k * 4, torch.full_like(lo, PAD_LO)).to(torch.int32))
            else:
                hi = slot * (hr * slot_bytes) + (dy * hz + dz) * slot_bytes + k * elem_bytes
                hi = torch.where(valid, hi, torch.full_like(hi, PAD_HI))
                out.append(torch.stack((lo, hi), -1).to(torch.int32))
    return torch.stack(out, 0).contiguous()


def workgroup_classes(plan: LatticePlan, ty: int, tz: int, nseg: int) -> torch.Tensor:
    """[workgroups][nloc] uint8: the classes that occur among the rows of each workgroup (0xff = unused), workgroup order
    ((item·nseg + seg)·tiles_y + tile_y)·tiles_z + tile_z as in the kernel.  A workgroup keeps only these records in LDS."""
    nb, nx, ny, nz = plan.nb, plan.nx, plan.ny, plan.nz
    n = plan.n_rows
    dev = plan.rcls.device
    seg_len = -(-nx // nseg)
    tiles_y, tiles_z = -(-ny // ty), -(-nz // tz)
    r = torch.arange(n, device=dev, dtype=torch.int64)
    z = r % nz
    y = torch.div(r, nz, rounding_mode="floor") % ny
    xx = torch.div(r, ny * nz, rounding_mode="floor")
    x = xx % nx
    item = torch.div(xx, nx, rounding_mode="floor")
    blk = ((item * nseg + torch.div(x, seg_len, rounding_mode="floor")) * tiles_y + torch.div(y, ty, rounding_mode="floor")) * tiles_z \
        + torch.div(z, tz, rounding_mode="floor")
    nblocks = nb * nseg * tiles_y * tiles_z
    pairs = torch.unique(blk * 256 + plan.rcls[:n].to(torch.int64))          # sorted: by workgroup, then class
    pb = torch.div(pairs, 256, rounding_mode="floor")
    counts = torch.bincount(pb, minlength=nblocks)
    nloc = int(counts.max())
    start = torch.zeros(nblocks + 1, dtype=torch.int64, device=dev)
    start[1:] = torch.cumsum(counts, 0)
    within = torch.arange(pairs.numel(), device=dev) - start[pb]
    wl = torch.full((nblocks, nloc), 0xFF, dtype=torch.uint8, device=dev)
    wl[pb, within] = (pairs % 256).to(torch.uint8)
    return wl.contiguous()


# ---- launch configuration ------------------------------------------------------------------------------------

_NLOC_GUESS = (8, 27)     # classes a workgroup meets at least (stored-order / transposed walk): optimistic LDS estimate for
                          # ranking; config_for checks the real lists and falls back to the next candidate
_CFG_ENV = os.environ.get("TSGU_LATTICE_CFG", "")   # "ty,tz,nseg,threads[,ring[,chunks per lane]]" overrides the choice (experiments)


def _candidates(limit: int):
    for ty in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32):
        for tz in (4, 5, 6, 8, 10, 12, 16, 20, 24, 32, 50, 64):
            if ty * tz <= limit:
                yield ty, tz


def choose_config(plan: LatticePlan, mode: int, vtype: int, p: int, elem_bytes: int, lds_bytes_fn) -> Optional[Tuple[int, int, int, int, int]]:
    """(ty, tz, nseg, threads, ring) for `plan`: the tile with the best modelled throughput that fits the kernels' limits."""
    ranked = rank_configs(plan, mode, vtype, p, elem_bytes, lds_bytes_fn)
    return ranked[0] if ranked else None


def rank_configs(plan: LatticePlan, mode: int, vtype: int, p: int, elem_bytes: int, lds_bytes_fn, keep: int = 6):
    """Candidate launch configurations, best modelled throughput first."""
    if _CFG_ENV:
        v = [int(t) for t in _CFG_ENV.split(",")]
        return [(v[0], v[1], min(v[2], plan.nx), v[3], (v[4] if len(v) > 4 else 4), (v[5] if len(v) > 5 else 1))]
    cl = p * elem_bytes // 16
    found = {}
    alpha = 0.35 if plan.kind == 0 else 0.8       # what a halo row costs relative to an own row (the transposed walk also stages its values)
    for threads in (256, 512, 1024):
        rpp = threads // cl
        for ty, tz in _candidates(rpp):           # one row per lane group and plane step
            if ty > plan.ny or tz > plan.nz or (plan.ny == 1 and ty != 1):
                continue
            if elem_bytes == 8 and tz % 4 and plan.nz >= 8:
                continue      # fp64: a wave's rows inside one z-line (measured at C2: 4x8 226 / 335 / 281 us, 5x6 263 / 368 / 292 us)
            ring = 4
            lds = lds_bytes_fn(mode, vtype, p, ty, tz, plan.ry, plan.rz, min(plan.ncls, _NLOC_GUESS[plan.kind]), plan.recw, threads, ring)
            if lds <= 0:
                continue
            per_cu = min(160 * 1024 // lds, 2048 // threads)
            if per_cu < 1:
                continue
            nr = ty * tz
            tiles = -(-plan.ny // ty) * -(-plan.nz // tz)
            lane_util = (plan.ny * plan.nz) / (tiles * rpp)
            halo = (ty + 2 * plan.ry) * (tz + 2 * plan.rz) / nr
            slots = num_cu(getattr(getattr(plan, 'rcls', None), 'device', None)) * per_cu
            # measured (C2, fp32): two independent workgroups per CU overlap each other's barrier / DMA waits (1.0); one
            # workgroup of 16 waves is ~15 % slower; 8 waves per CU cannot hide the LDS latency (~1.5x)
            waves = per_cu * threads // 64
            hide = 1.5 if waves < 16 else (1.0 if per_cu >= 2 else 1.15)
            base = plan.nb * tiles
            # x segments: fill the chip's workgroup slots evenly; every segment re-reads two halo planes
            for nseg in range(1, min(plan.nx, 64) + 1):
                seg_len = -(-plan.nx // nseg)
                if seg_len * (nseg - 1) >= plan.nx:
                    continue
                nwg = base * nseg
                fill = nwg / (-(-nwg // slots) * slots)
                cost = (1.0 / lane_util) * (1.0 + alpha * (halo - 1.0)) * (1.0 + 2.0 / seg_len) / fill * hide
                key = (ty, tz, threads)
                if key not in found or cost < found[key][0]:
                    found[key] = (cost, (ty, tz, nseg, threads, ring, 1))
    if mode == _MODE_SPMMT and elem_bytes == 2:
        # bf16 transposed walk: its ring (dense rows + value rows of the halo) and per-phase record tables leave room for ONE
        # workgroup per CU, and four waves cannot cover the LDS latency of their own reads.  The same tile run by more threads —
        # only every second / fourth row group busy in the products, all of them moving the ring — is faster (measured at C5,
        # 64 items, 8x16 tile: 256 threads 561 us, 512 threads 508 us, 1024 threads 474 us)
        for (ty, tz, threads), (cost, c) in list(found.items()):
            if threads != 256:
                continue
            for more, gain in ((512, 0.92), (1024, 0.85)):
                lds = lds_bytes_fn(mode, vtype, p, ty, tz, plan.ry, plan.rz, min(plan.ncls, _NLOC_GUESS[plan.kind]), plan.recw, more, c[4])
                if lds > 0 and 160 * 1024 // lds < 2:
                    found[(ty, tz, more)] = (cost * gain, (ty, tz, c[2], more, c[4], c[5]))
    return [c for _, c in sorted(found.values())[:keep]]


class _LatticePlanStruct(ctypes.Structure):
    """``tsgu_lattice_plan`` of include/tsgu_hip.h."""

    _fields_ = [(k, ctypes.c_int32) for k in ("kind", "nb", "nx", "ny", "nz", "ry", "rz", "ncls", "recw", "nloc", "uniform_len",
                                               "ty", "tz", "nseg", "threads", "ring", "chunks_per_lane")] + [(k, ctypes.c_void_p) for k in ("rec", "lens", "rcls", "rstart", "wlist")]


def build_config(plan: LatticePlan, cand, mode: int, vtype: int, p: int, elem_bytes: int, lds_bytes_fn, be=None) -> Optional[LatticeConfig]:
    """The launch configuration `cand` = (ty, tz, nseg, threads, ring, chunks per lane) with its record tables on the device, or
    None when the workgroups of this tiling meet more classes than fit."""
    ty, tz, nseg, threads, ring, cpl = cand
    wlist = workgroup_classes_hip(plan, ty, tz, nseg, be) if be is not None else workgroup_classes(plan, ty, tz, nseg)
    nloc = wlist.size(1)
    lds = lds_bytes_fn(mode, vtype, p, ty, tz, plan.ry, plan.rz, nloc, plan.recw, threads, ring, cpl)
    if lds <= 0:
        return None
    slot = (plan.recw * (max(4, elem_bytes if elem_bytes == 8 else 4) if mode == _MODE_SDDMM else elem_bytes) + 15) // 16 * 16
    if slot % 64 == 0:
        slot += 16                 # as lat_layout (csrc/lattice_impl.h): value rows must not share four banks
    if PACKED_T and mode == _MODE_SPMMT and (p * elem_bytes) % 128 == 0:
        slot = p * elem_bytes      # packed records: the value ring has the pitch of the dense ring
    cfg = LatticeConfig()
    cfg.ty, cfg.tz, cfg.nseg, cfg.threads, cfg.lds_bytes = ty, tz, nseg, threads, lds
    cfg.wlist, cfg.nloc, cfg.ring, cfg.cpl = wlist, nloc, ring, cpl
    cfg.uses, cfg.tuned = 0, False
    cfg.rec = records(plan, ty, tz, p * elem_bytes, slot, ring, elem_bytes).to(plan.rcls.device)
    cfg.struct = _LatticePlanStruct(plan.kind, plan.nb, plan.nx, plan.ny, plan.nz, plan.ry, plan.rz, plan.ncls, plan.recw,
                                    nloc, plan.uniform_len, ty, tz, nseg, threads, ring, cpl, cfg.rec.data_ptr(), plan.lens.data_ptr(),
                                    plan.rcls.data_ptr(), plan.rstart.data_ptr(), wlist.data_ptr())
    cfg.struct_addr = ctypes.addressof(cfg.struct)
    return cfg


def config_for(plan: LatticePlan, mode: int, vtype: int, p: int, elem_bytes: int, lds_bytes_fn, be=None) -> Optional[LatticeConfig]:
    """Cached launch configuration (tile choice + record tables on the device + ctypes image) of a plan: the best-ranked
    candidate that fits (a pattern that keeps coming back gets the measured choice, `tune_config`)."""
    key = (mode, vtype, p)
    cfg = plan._cfg.get(key)
    if cfg is None and key not in plan._cfg:
        for cand in rank_configs(plan, mode, vtype, p, elem_bytes, lds_bytes_fn):
            cfg = build_config(plan, cand, mode, vtype, p, elem_bytes, lds_bytes_fn, be)
            if cfg is not None:
                break
        plan._cfg[key] = cfg
    return cfg


# The ranking is a model fitted at C2 (fp32, 32 columns); other operands (C5: bf16, 16 columns) have their best tile elsewhere
# (measured: forward 8x16 / 256 threads 300 us against the model's 16x16 / 512 at 338 us; SDDMM 16x32 / 1024 threads 285 us
# against 330 us).  A pattern that keeps coming back is therefore MEASURED once: a few launches of the best-ranked candidates
# of every workgroup size on the caller's operands.  Every configuration sums a row in the same order, so the choice does not
# change a result bit.
TUNE = os.environ.get("TSGU_LATTICE_TUNE", "1") == "1"
TUNE_AFTER_USES = int(os.environ.get("TSGU_LATTICE_TUNE_AFTER", "3"))
TUNE_PER_SIZE = 3
TUNE_MARGIN = 0.93    # a measured candidate replaces the ranked configuration only when it is at least 7 % faster in the trial
TUNE_LOG = []        # (plan kind, mode, vtype, p, [(candidate, ms)], chosen) of every measured choice (diagnostics, tests)


def tune_candidates(plan: LatticePlan, mode: int, vtype: int, p: int, elem_bytes: int, lds_bytes_fn):
    """The TUNE_PER_SIZE best-ranked candidates of every workgroup size, best-ranked first."""
    if _CFG_ENV:
        return []
    ranked = rank_configs(plan, mode, vtype, p, elem_bytes, lds_bytes_fn, keep=1 << 30)
    taken, per = [], {}
    for c in ranked:
        if per.get(c[3], 0) < TUNE_PER_SIZE:
            per[c[3]] = per.get(c[3], 0) + 1
            taken.append(c)
    return taken


def tune_config(plan: LatticePlan, mode: int, vtype: int, p: int, elem_bytes: int, lds_bytes_fn, be, time_ms) -> Optional[LatticeConfig]:
    """Replace the cached configuration of (mode, vtype, p) by the fastest of `tune_candidates`; `time_ms(cfg)` launches the
    kernel on the caller's operands and returns milliseconds per launch.  Returns the chosen configuration."""
    key = (mode, vtype, p)
    cur = plan._cfg.get(key)
    if cur is None:
        return None
    cur.tuned = True
    tried = []
    best, best_ms, cur_ms = cur, None, None
    for cand in tune_candidates(plan, mode, vtype, p, elem_bytes, lds_bytes_fn):
        same = cand == (cur.ty, cur.tz, cur.nseg, cur.threads, cur.ring, cur.cpl)
        cfg = cur if same else build_config(plan, cand, mode, vtype, p, elem_bytes, lds_bytes_fn, be)
        if cfg is None:
            continue
        ms = time_ms(cfg)
        tried.append((cand, ms))
        if same:
            cur_ms = ms
        if best_ms is None or ms < best_ms:
            best, best_ms = cfg, ms
    # the ranked configuration stays unless a candidate beats it by a margin the trial can resolve: at C2-like lattices the model's
    # first choice is within a few per cent of the best, and a choice made on timing noise cost up to 20 % of a step (round 4)
    if cur_ms is not None and best is not cur and best_ms > TUNE_MARGIN * cur_ms:
        best = cur
    best.tuned = True
    best.uses = cur.uses
    plan._cfg[key] = best
    TUNE_LOG.append((plan.kind, mode, vtype, p, tried, (best.ty, best.tz, best.nseg, best.threads, best.ring, best.cpl)))
    return best


# ---- plane-march kernels (csrc/march_impl.h): full periodic box stencils --------------------------------------------
ENABLE_MARCH = os.environ.get("TSGU_ENABLE_MARCH", "1") == "1"
ENABLE_MARCH_RAW = os.environ.get("TSGU_MARCH_RAW", "1") != "0"      # periodic whole box: raw value rows (no gathers for the rows that wrap)
_MARCH_CFG_ENV = os.environ.get("TSGU_MARCH_CFG", "")   # "ty,tz,nseg,threads" overrides the choice (experiments)
# … and per product (TSGU_MARCH_CFG_FWD / _SDDMM / _SPMMT): in-step probes of one kernel's configuration with the others unchanged
_MARCH_CFG_MODE_ENV = {m: os.environ.get("TSGU_MARCH_CFG_" + n, "") for m, n in ((0, "FWD"), (1, "SDDMM"), (2, "SPMMT"))}
MARCH_TAPS = 9
MARCH_MAX_CLASSES = 64
_MARCH_WAVES_PER_CU = {0: 20, 1: 20, 2: 16}     # resident waves per CU the segment count is planned for
# SDDMM: x-segments = this factor times the ranked count (segments of at least 8 planes).  Measured INSIDE the step (rocprofv3 kernel
# statistics of the C2 step, one box, us): 3 segments (ranked) 88.0, 5: 85.6, 6: 83.4-84.2, 7: 83.4, 8: 86.1, 10: 84.8, 12: 89.7 — the
# forward at the same counts gets slower (74.8 -> 78.8 at 6).  A trial launch behind a device copy (the sweeps' way of choosing) ranks
# them the other way round (104.7 against 108.5 us): in the step the SDDMM follows the forward and finds its operands in MALL / L2.
# Only for rows of one length on the whole box (periodic lattices): same box, in the step, factor 1 / 2: periodic 84.6 / 81.1 us,
# truncated box 87.3 / 87.5, its lower triangle 61.0 / 65.1.
# (late round 6: with the select-free eight-lane sums the SDDMM issues a quarter fewer instructions and the ranked count is the faster one
# again — same box, in the step, three alternations: 3 segments 78.5-79.1 us, 6 segments 80.9-81.4 us — so the factor defaults to 1)
MARCH_SDDMM_SEGMENT_FACTOR = int(os.environ.get("TSGU_MARCH_SDDMM_SEGMENT_FACTOR", "1"))
# workgroup sizes in order of preference: the first that fits the lattice is taken (measured at C2, same box, us:
# forward 4x8/256: 80.8-85.5, 8x8/512: 88.1;  SDDMM 8x8/512: 87.4, 4x8/256: 94.6-101.7;  transposed 8x8/512: 102.4, 4x8/256: 99.5-103.0)
# round 4 (three alternations per configuration in one process, C2, us): SDDMM 4x8/256 80.1-83.0 against 8x8/512 84.4-97.5 — the
# smaller workgroups fill the SIMDs' register files evenly (80 registers: six 256-thread workgroups or three of 512 per CU)
_MARCH_THREADS = {0: (256, 512), 1: (256, 512), 2: (512, 256), 3: (256,)}
_MARCH_HALO_COST = {0: 0.1, 1: 0.1, 2: 0.8, 3: 0.8}     # what a halo row costs relative to an own row


class MarchTables:
    """What the plane-march kernels need besides the lattice plan's `rcls` / `rstart`: the displacement set, the canonical class
    and the per-class map canonical slot -> stored position (include/tsgu_hip.h, tsgu_march_plan)."""

    __slots__ = ("ident", "taps", "mask", "periodic", "full", "kidx_host", "kidx", "_cfg", "_line_ok")


class MarchConfig:
    """One launch configuration of the plane-march kernels (quacks like LatticeConfig where bench.py / tests look).
    `col_tile`: columns per launch (operands wider than 64 columns run as tiles of 64)."""

    __slots__ = ("mode", "ty", "tz", "nseg", "threads", "lds_bytes", "struct", "struct_addr", "ring", "cpl", "nloc", "tables", "col_tile")
    march = True


class _MarchPlanStruct(ctypes.Structure):
    """``tsgu_march_plan`` of include/tsgu_hip.h."""

    _fields_ = ([(k, ctypes.c_int32) for k in ("nb", "nx", "ny", "nz", "ry", "rz", "ntap")] + [("tap_dy", ctypes.c_int32 * 9), ("tap_dz", ctypes.c_int32 * 9)]
                + [(k, ctypes.c_int32) for k in ("ncls", "ident", "ty", "tz", "nseg", "threads")] + [("mask", ctypes.c_uint32)]
                + [(k, ctypes.c_int32) for k in ("periodic", "uniform_len")] + [(k, ctypes.c_void_p) for k in ("kidx", "rcls", "rstart")])


MARCH_FULL = (1 << 27) - 1


def march_tables(plan: LatticePlan) -> Optional[MarchTables]:
    """MarchTables of a stored-order lattice plan that meets the plane-march condition (`plan.box`: every row holds exactly the
    displacements of one subset of the 3 x 3 x 3 box whose neighbour exists — periodic or truncated 27- / 7-point stencils,
    triangular parts of truncated ones … on a lattice of at least 3 points per dimension), else None.  Host work on the
    [classes][28] code table only."""
    if plan._march is not False:
        return plan._march
    plan._march = None
    if plan.kind != 0 or plan.box is None or plan.ncls > MARCH_MAX_CLASSES or min(plan.nx, plan.ny, plan.nz) < 3:
        return None
    mask, periodic = plan.box

    def slot_of(code):
        return (code // 25) * 9 + ((code // 5) % 5 - 1) * 3 + code % 5 - 1

    canon = [s for s in range(27) if mask >> s & 1]                     # ascending (dx, dy, dz)
    ident = None
    kidx = [[0xFF] * 32 for _ in range(plan.ncls)]
    for c, row in enumerate(plan.codes.tolist()):
        slots = [slot_of(code) for code in row if code >= 0]
        if len(set(slots)) != len(slots) or len(slots) > 27:
            return None
        if slots == canon:
            ident = c
        for k, sl in enumerate(slots):
            kidx[c][sl] = k
        kidx[c][31] = len(slots)
    if ident is None:
        return None
    mt = MarchTables()
    mt.ident, mt.mask, mt.periodic, mt.full = ident, mask, periodic, mask == MARCH_FULL
    mt.taps = [(dy, dz) for dy in (-1, 0, 1) for dz in (-1, 0, 1)]
    mt.kidx_host = torch.tensor(kidx, dtype=torch.uint8)
    mt.kidx = mt.kidx_host.to(plan.rcls.device)
    mt._cfg = {}
    plan._march = mt
    return mt


def march_config_for(plan: LatticePlan, mode: int, vtype: int, p: int, lds_bytes_fn, supported_fn=None) -> Optional[MarchConfig]:
    """Cached launch configuration of the plane-march kernels for a stored-order plan, or None (pattern / operands not covered;
    `supported_fn(mode, mask, uniform_len, threads)`: which kernels the library holds — the whole box: all products; triangular
    halves: the SDDMM)."""
    if not ENABLE_MARCH or vtype != 0 or not (p in (16, 32, 64) or (p > 64 and p % 64 == 0 and p <= 1024)):
        return None
    if p == 16 and mode == 0 and not _MARCH_CFG_ENV:
        return None      # 16 columns, forward: the general sweep is faster (measured at C2's lattice: 46 against 58 us; SDDMM 89 / 77, transposed 86 / 71)
    mt = march_tables(plan)
    if mt is None:
        return None
    if mt.full and not plan.uniform_len and 36 * plan.ny * plan.nz >= 1 << 24:
        return None      # truncated box: the kernels' row-start arithmetic uses 24-bit multiplies (planes under ~466 000 points)
    key = (mode, p)
    if key in mt._cfg:
        return mt._cfg[key]
    pt = min(p, 64)                 # columns per launch
    cl = pt // 4
    best = None
    pinned = _MARCH_CFG_MODE_ENV.get(mode) or _MARCH_CFG_ENV
    if pinned:
        v = [int(t) for t in pinned.split(",")]
        cands = [(v[0], v[1], min(v[2], plan.nx), v[3])]
        if supported_fn is not None and not supported_fn(mode, mt.mask, plan.uniform_len, v[3]):
            cands = []
    else:
        # a wave = 64 / cl consecutive rows of one z-line (conflict-free LDS row reads): tz = 8 rows (p = 32) or a multiple
        cands = []
        for threads in _MARCH_THREADS[mode]:
            if supported_fn is not None and not supported_fn(mode, mt.mask, plan.uniform_len, threads):
                continue
            rpp = threads // cl
            tz = 8
            if rpp // tz >= 1 and rpp // tz <= plan.ny and tz <= plan.nz:
                cands.append((rpp // tz, tz, 0, threads))
    for ty, tz, nseg, threads in cands:
        lds = lds_bytes_fn(mode, vtype, pt, ty, tz, 1, 1, plan.ncls, threads)
        if lds <= 0:
            continue
        per_cu = max(1, min(160 * 1024 // lds, _MARCH_WAVES_PER_CU[mode] * 64 // threads))
        slots = num_cu(getattr(getattr(plan, 'rcls', None), 'device', None)) * per_cu
        tiles = -(-plan.ny // ty) * -(-plan.nz // tz)
        base = plan.nb * tiles
        util = (plan.ny * plan.nz) / (tiles * ty * tz)
        halo = (ty + 2) * (tz + 2) / (ty * tz)
        util /= 1.0 + _MARCH_HALO_COST[mode] * (halo - 1.0)
        choices = [nseg] if nseg else range(1, min(plan.nx, 64) + 1)
        for ns_ in choices:
            seg_len = -(-plan.nx // ns_)
            if seg_len * (ns_ - 1) >= plan.nx:
                continue
            nwg = base * ns_
            # a workgroup marches seg_len + 2 source planes (+ ~3 steps of prologue); workgroups run in rounds of `slots`
            cost = -(-nwg // slots) * (seg_len + 5) / util
            if best is None or cost < best[0]:
                best = (cost, ty, tz, ns_, threads, lds)
        if best is not None:
            break
    cfg = None
    if best is not None:
        _, ty, tz, nseg, threads, lds = best
        if mode == 1 and not pinned and MARCH_SDDMM_SEGMENT_FACTOR > 1 and mt.full and plan.uniform_len:
            # rows of one length on the whole box (the kernel whose stores leave as aligned pieces of a wave's run): faster on MORE,
            # shorter x-segments than the one-round ranking picks (see MARCH_SDDMM_SEGMENT_FACTOR)
            want = nseg * MARCH_SDDMM_SEGMENT_FACTOR
            if want <= 64 and -(-plan.nx // want) >= 8 and -(-plan.nx // want) * (want - 1) < plan.nx:
                nseg = want
        cfg = MarchConfig()
        cfg.mode, cfg.ty, cfg.tz, cfg.nseg, cfg.threads, cfg.lds_bytes = mode, ty, tz, nseg, threads, lds
        cfg.ring, cfg.cpl, cfg.nloc, cfg.tables, cfg.col_tile = 2, 1, plan.ncls, mt, pt
        dy = (ctypes.c_int32 * 9)(*[t[0] for t in mt.taps])
        dz = (ctypes.c_int32 * 9)(*[t[1] for t in mt.taps])
        # periodic whole box with sorted columns: the kernels may stage value rows RAW (bit 3 of `periodic`: the stored position of a
        # displacement is 9·rank_x + 3·rank_y + rank_z — checked once per pattern, not while a stream is being captured)
        periodic = mt.periodic
        if (ENABLE_MARCH_RAW and plan.rcls.is_cuda and mt.full and mt.periodic == 7 and plan.uniform_len == 27
                and (getattr(mt, "_line_ok", None) is not None or not torch.cuda.is_current_stream_capturing()) and linemarch_ok(plan, mt)):
            periodic |= 8
        cfg.struct = _MarchPlanStruct(plan.nb, plan.nx, plan.ny, plan.nz, 1, 1, MARCH_TAPS, dy, dz, plan.ncls, mt.ident,
                                      ty, tz, nseg, threads, mt.mask, periodic, plan.uniform_len, mt.kidx.data_ptr(),
                                      plan.rcls.data_ptr(), plan.rstart.data_ptr())
        cfg.struct_addr = ctypes.addressof(cfg.struct)
    mt._cfg[key] = cfg
    return cfg


# ---- whole-line march (csrc/linemarch_impl.h): bf16, 16 columns, periodic 27-point box --------------------------------------
ENABLE_LINEMARCH = os.environ.get("TSGU_ENABLE_LINEMARCH", "1") != "0"
LINEMARCH_MODES = tuple(int(t) for t in os.environ.get("TSGU_LINEMARCH_MODES", "0,1,2").split(",") if t)     # products it takes: 0 A·B, 1 SDDMM, 2 Aᵀ·G
_LINEMARCH_CFG_ENV = os.environ.get("TSGU_LINEMARCH_CFG", "")      # "ty,nseg" pins the tile height and the x-segments (experiments)
_LINE_RANK = ((2, 0, 1), (0, 1, 2), (1, 2, 0))      # rank of the neighbour at d = -1, 0, +1 for a point at the lower face / inside / at the upper face


def linemarch_ok(plan: LatticePlan, mt: MarchTables) -> bool:
    """The kernels compute the stored position of displacement (dx, dy, dz) in the row at (x, y, z) as 9·rank_x + 3·rank_y + rank_z
    (sorted columns on a periodic lattice) instead of reading the plan's class tables: true when that arithmetic reproduces the
    tables for every class AND every row's class is the one of its position.  Once per pattern (one device reduction, one host read)."""
    hit = getattr(mt, "_line_ok", None)
    if hit is not None:
        return hit
    ok = False
    if plan.ncls == 27 and mt.full and mt.periodic == 7 and plan.uniform_len == 27:
        kidx = mt.kidx_host[:, :27].tolist()
        cls_of = {}
        for sx in range(3):
            for sy in range(3):
                for sz in range(3):
                    want = [9 * _LINE_RANK[sx][dx] + 3 * _LINE_RANK[sy][dy] + _LINE_RANK[sz][dz] for dx in range(3) for dy in range(3) for dz in range(3)]
                    if want in kidx:
                        cls_of[(sx, sy, sz)] = kidx.index(want)
        if len(cls_of) == 27 and len(set(cls_of.values())) == 27:
            dev = plan.rcls.device

            def state(n):
                s = torch.ones(n, dtype=torch.long, device=dev)
                s[0], s[n - 1] = 0, 2
                return s

            tab = torch.tensor([[[cls_of[(a, b, c)] for c in range(3)] for b in range(3)] for a in range(3)], dtype=plan.rcls.dtype, device=dev)
            want = tab[state(plan.nx)][:, state(plan.ny)][:, :, state(plan.nz)]
            n_rows = plan.nb * plan.nx * plan.ny * plan.nz
            ok = bool((plan.rcls[:n_rows].view(plan.nb, plan.nx, plan.ny, plan.nz) == want).all().item())
    mt._line_ok = ok
    return ok


def linemarch_config_for(plan: LatticePlan, mode: int, vtype: int, p: int, lds_bytes_fn) -> Optional[MarchConfig]:
    """Launch configuration of the whole-line march for a stored-order plan, or None: bf16, 16 columns, a product (mode 0 A·B, 1 SDDMM,
    2 Aᵀ·G) of a periodic 27-point box stencil with sorted columns whose z-lines have one of the lengths the kernels are compiled for
    (8 / 16 / 32 / 64) and whose tile height (threads / (2·nz) lines, 512 threads preferred) divides ny."""
    if not ENABLE_LINEMARCH or not ENABLE_MARCH or vtype != 2 or p != 16 or mode not in LINEMARCH_MODES or not plan.rcls.is_cuda:
        return None
    mt = march_tables(plan)
    if mt is None or not mt.full or mt.periodic != 7 or plan.uniform_len != 27 or plan.nz not in (8, 16, 32, 64):      # (the kernels are compiled per line length)
        return None
    key = ("line", mode, p)
    if key in mt._cfg:
        return mt._cfg[key]
    cfg = None
    if torch.cuda.is_current_stream_capturing():
        return None          # (linemarch_ok reads a device reduction)
    if linemarch_ok(plan, mt):
        best = None
        pinned = [int(t) for t in _LINEMARCH_CFG_ENV.split(",")] if _LINEMARCH_CFG_ENV else None
        for threads in (512, 1024, 256):
            if threads % (2 * plan.nz):
                continue
            ty = threads // (2 * plan.nz)
            if pinned:
                if ty != pinned[0]:
                    continue
            if ty > plan.ny or plan.ny % ty:
                continue
            lds = lds_bytes_fn(mode, vtype, p, ty, plan.nz, 1, 1, plan.ncls, threads)
            if lds <= 0:
                continue
            per_cu = max(1, min(160 * 1024 // lds, 2048 // threads))
            slots = num_cu(plan.rcls.device) * per_cu
            base = plan.nb * (plan.ny // ty)
            halo = (ty + 2) / ty
            for ns_ in ([pinned[1]] if pinned else range(1, min(plan.nx, 64) + 1)):
                seg_len = -(-plan.nx // ns_)
                if seg_len * (ns_ - 1) >= plan.nx:
                    continue
                cost = -(-base * ns_ // slots) * (seg_len + 3) * threads * halo
                if best is None or cost < best[0]:
                    best = (cost, ty, ns_, threads, lds)
            if best is not None and not pinned:
                break
        if best is not None:
            _, ty, nseg, threads, lds = best
            cfg = MarchConfig()
            cfg.mode, cfg.ty, cfg.tz, cfg.nseg, cfg.threads, cfg.lds_bytes = mode, ty, plan.nz, nseg, threads, lds
            cfg.ring, cfg.cpl, cfg.nloc, cfg.tables, cfg.col_tile = 2, 1, plan.ncls, mt, p
            dy = (ctypes.c_int32 * 9)(*[t[0] for t in mt.taps])
            dz = (ctypes.c_int32 * 9)(*[t[1] for t in mt.taps])
            cfg.struct = _MarchPlanStruct(plan.nb, plan.nx, plan.ny, plan.nz, 1, 1, MARCH_TAPS, dy, dz, plan.ncls, mt.ident,
                                          ty, plan.nz, nseg, threads, mt.mask, mt.periodic, plan.uniform_len, mt.kidx.data_ptr(),
                                          plan.rcls.data_ptr(), plan.rstart.data_ptr())
            cfg.struct_addr = ctypes.addressof(cfg.struct)
    mt._cfg[key] = cfg
    return cfg
