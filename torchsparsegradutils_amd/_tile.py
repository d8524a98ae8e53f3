"""Plans of the row-block tile kernels (csrc/tile_impl.h, ``tsgu_tile_plan`` in include/tsgu_hip.h).

A plan is value independent and built once per sparsity pattern (cached with it): for every block of ``rows_per_block`` consecutive
rows the ascending list of DISTINCT columns its rows reference (the block's tile, copied to LDS by the kernel one block ahead of the
walk) and, per stored entry, one byte naming the entry's column inside that list.  The reference re-derives its structure on every
call (``repeat_interleave`` of the row pointer, the CSC→CSR sort inside ``torch.sparse.mm(A.t(), ·)``; sparse_matmul.py:186-192,229).

A pattern qualifies when every block's tile fits the kernel's LDS budget (``max_union`` rows) and the tiles are worth staging:
on average an entry must share its dense row with others of its block (``REUSE_MIN``) — a random band matrix, whose rows share
nothing, stays on the gather kernels.
"""

from __future__ import annotations

import ctypes
from typing import Optional

import torch

# stored entries per distinct dense row of a block below which a tile is not worth staging (a 27-point mesh numbered in 4^3 bricks: 8.0;
# a banded random matrix: ~1.0)
REUSE_MIN = 2.0
# value chunks per block the kernels can fetch (csrc/tile_impl.h: kTileCMax)
MAX_CHUNKS = 1024
# bytes of a tile row in LDS (a column tile of 32 fp32 columns): entry records hold the byte offset of their tile row
TILE_ROW_BYTES = 128


def _owned(t: torch.Tensor, source: torch.Tensor) -> torch.Tensor:
    """`t`, or a copy when it IS the caller's tensor (an int32 contiguous row pointer converts to itself): a plan never holds the
    caller's index tensors — the pattern cache drops an entry when they die and may keep the plans adoptable beyond that."""
    return t.clone() if t.untyped_storage()._cdata == source.untyped_storage()._cdata else t


class TilePlan:
    """Device arrays of one ``tsgu_tile_plan`` (+ its ctypes image, cached by _backend)."""

    __slots__ = ("n_rows", "n_cols", "nnz", "n_blocks", "rows_per_block", "max_union", "max_entries", "desc", "ucol", "lidx", "rptr", "cpos",
                 "cslot", "ent", "xrow", "reuse", "_cstruct")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))

    def plan_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in (self.desc, self.ucol, self.lidx, self.rptr, self.cpos, self.cslot, self.ent, self.xrow) if t is not None)


def build_tile_plan(crow: torch.Tensor, col: torch.Tensor, n_rows: int, n_cols: int, rows_per_block: int, max_union: int, max_entries: int,
                    perm: Optional[torch.Tensor] = None, rows: Optional[torch.Tensor] = None, reuse_min: float = REUSE_MIN) -> Optional[TilePlan]:
    """TilePlan of the 2-D pattern (crow, col) or None when it does not qualify.  Tensor ops on the pattern's device: one sort of
    (block, column) keys + a few scans; the only host reads are the two limits and the reuse figure."""
    nnz = col.numel()
    if n_rows <= 0 or nnz <= 0 or nnz >= 2**31 or n_rows >= 2**31 or n_cols >= 2**31 or max_union * TILE_ROW_BYTES >= 2**15:
        return None
    dev = col.device
    R = rows_per_block
    nb = (n_rows + R - 1) // R
    crow64 = crow.to(torch.int64)
    if rows is None:
        rows = torch.repeat_interleave(torch.arange(n_rows, device=dev, dtype=torch.int64), crow64[1:] - crow64[:-1], output_size=nnz)
    blk = rows.to(torch.int64) // R
    e0 = crow64[0:n_rows:R]                                             # first entry of every block
    e1 = torch.cat((e0[1:], crow64[n_rows:n_rows + 1]))
    if int((e1 - e0).max()) > max_entries:
        return None
    key = blk * n_cols + col.to(torch.int64)
    ukey, inv = torch.unique(key, sorted=True, return_inverse=True)
    ublk = ukey // n_cols
    cnt = torch.bincount(ublk, minlength=nb)                            # distinct columns per block
    if int(cnt.max()) > max_union:
        return None
    reuse = nnz / max(int(ukey.numel()), 1)
    if reuse < reuse_min:
        return None
    first = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    first[1:] = torch.cumsum(cnt, 0)                                    # unpadded offsets of the blocks' lists in ukey
    padded = (cnt + 7) // 8 * 8                                         # lists padded to whole 8-row DMA instructions
    u0 = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    u0[1:] = torch.cumsum(padded, 0)
    total = int(u0[-1])
    if total >= 2**31:
        return None
    # ucol: every padded slot holds the block's LAST column (a harmless duplicate request), then the real lists are scattered in
    last_col = (ukey % n_cols)[first[1:] - 1]
    ucol = torch.repeat_interleave(last_col, padded, output_size=total)
    pos = u0[ublk] + (torch.arange(ukey.numel(), device=dev, dtype=torch.int64) - first[ublk])
    ucol[pos] = ukey % n_cols
    lidx = torch.zeros(nnz + 16, dtype=torch.uint8, device=dev)
    lidx[:nnz] = (inv - first[blk]).to(torch.uint8)
    # the forward / Aᵀ·G walk reads RECORDS: per row whole rounds of eight 16-bit tile offsets (position in the tile · the bytes of a
    # tile row), the last round padded with the offset of the kernel's zero row (max_union · row bytes)
    lens = crow64[1:n_rows + 1] - crow64[:n_rows]
    rounds = (lens + 7) // 8
    rcum = torch.zeros(n_rows + 1, dtype=torch.int64, device=dev)
    rcum[1:] = torch.cumsum(rounds, 0)
    x0 = rcum[0:n_rows:R]                                                # first record of every block
    x1 = torch.cat((x0[1:], rcum[n_rows:n_rows + 1]))
    nrec = int(rcum[-1])
    xrow = torch.zeros(nb * R, dtype=torch.int64, device=dev)
    xrow[:n_rows] = rcum[:n_rows] - torch.repeat_interleave(x0, R, output_size=nb * R)[:n_rows]
    ent = torch.full((nrec + 8, 8), max_union * TILE_ROW_BYTES, dtype=torch.int64, device=dev)
    q = torch.arange(nnz, device=dev, dtype=torch.int64) - crow64[rows.to(torch.int64)]      # position of every entry inside its row
    ent[rcum[rows.to(torch.int64)] + q // 8, q % 8] = (inv - first[blk]) * TILE_ROW_BYTES
    desc = torch.zeros((nb + 4, 8), dtype=torch.int32, device=dev)
    desc[:nb, 6] = x0.to(torch.int32)
    desc[:nb, 7] = (x1 - x0).to(torch.int32)
    desc[:nb, 0] = u0[:-1].to(torch.int32)
    desc[:nb, 1] = padded.to(torch.int32)
    desc[:nb, 2] = e0.to(torch.int32)
    desc[:nb, 3] = (e1 - e0).to(torch.int32)
    cpos = cslot = None
    if perm is not None:
        # The walked pattern's values live elsewhere (a transposed pattern read through A's own value array): entry e of the walk is
        # val[perm[e]].  A block of a transposed mesh pattern draws RUNS of consecutive values (one run per source row that touches the
        # block, ~8 values each) from the value array.  The kernel fetches them as 16-byte CHUNKS in ascending source order — one load per
        # lane, neighbouring lanes on neighbouring chunks — and `cslot` says which entry of the block each of a chunk's four values is
        # (0xffff: a value that is not the block's).  Round 5 kept a 4-byte position + a 2-byte slot per ENTRY (6 bytes, 159 MB per
        # launch at N = 1e6); a chunk record is 12 bytes for ~3.3 entries.
        if nnz < 4:
            return None
        perm64 = perm.to(torch.int64)
        order = torch.argsort(blk * nnz + perm64)                        # (perm is a permutation: the keys are distinct; < 2^31 · 2^31)
        src = perm64[order]                                              # positions in the value array, ascending inside every block
        sblk = blk[order]
        slot = order - e0[sblk]                                          # the entry of its block every fetched value belongs to
        ar = torch.arange(nnz, device=dev, dtype=torch.int64)
        new_run = torch.ones(nnz, dtype=torch.bool, device=dev)
        new_run[1:] = (src[1:] != src[:-1] + 1) | (sblk[1:] != sblk[:-1])
        run_first = torch.cummax(torch.where(new_run, ar, torch.zeros_like(ar)), 0).values      # index of the first value of the run
        in_run = ar - run_first
        new_chunk = in_run % 4 == 0                                      # a run of L values: ceil(L / 4) chunks
        chunk = torch.cumsum(new_chunk.to(torch.int64), 0) - 1
        nchunks = int(chunk[-1]) + 1
        first_pos = src[new_chunk]                                       # (chunks are numbered in the order of their first values)
        start = torch.clamp(first_pos, max=nnz - 4)                      # a chunk never reads beyond the value array
        cs = torch.full((nchunks, 4), 0xFFFF, dtype=torch.int64, device=dev)
        cs[chunk, src - start[chunk]] = slot
        cblk = sblk[new_chunk]
        ccnt = torch.bincount(cblk, minlength=nb)
        if int(ccnt.max()) > MAX_CHUNKS:
            return None
        c0 = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
        c0[1:] = torch.cumsum(ccnt, 0)
        desc[:nb, 4] = c0[:-1].to(torch.int32)
        desc[:nb, 5] = ccnt.to(torch.int32)
        cpos = start.to(torch.int32).contiguous()
        cslot = cs.to(torch.int16).contiguous()                          # [chunks][4] uint16 bit patterns
    return TilePlan(n_rows=n_rows, n_cols=n_cols, nnz=nnz, n_blocks=nb, rows_per_block=R, max_union=max_union, max_entries=max_entries,
                    desc=desc.contiguous(), ucol=ucol.to(torch.int32).contiguous(), lidx=lidx, rptr=_owned(crow.to(torch.int32).contiguous(), crow),
                    cpos=cpos, cslot=cslot, ent=ent.to(torch.int16).contiguous(), xrow=xrow.to(torch.int16).contiguous(), reuse=reuse, _cstruct=None)


class TilePlanStruct(ctypes.Structure):
    """``tsgu_tile_plan`` of include/tsgu_hip.h."""

    _fields_ = [("n_rows", ctypes.c_int64), ("n_cols", ctypes.c_int64), ("nnz", ctypes.c_int64), ("n_blocks", ctypes.c_int64),
                ("rows_per_block", ctypes.c_int32), ("max_union", ctypes.c_int32), ("max_entries", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("desc", ctypes.c_void_p), ("ucol", ctypes.c_void_p), ("lidx", ctypes.c_void_p), ("rptr", ctypes.c_void_p), ("cpos", ctypes.c_void_p),
                ("cslot", ctypes.c_void_p), ("ent", ctypes.c_void_p), ("xrow", ctypes.c_void_p)]
